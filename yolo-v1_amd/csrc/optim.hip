// Fused optimizer step for gfx950 (HBM-bound, one pass over the 271.7 M fp32 parameters):
// global-gradient-norm clipping + Adam with L2 weight decay, i.e. what the reference's train step
// does with torch.nn.utils.clip_grad_norm_(max_norm=10) followed by optim.Adam(lr, weight_decay)
// (src/yolo/training/trainer.py:79-95, src/train.py:177-179) in ~10 separate multi-tensor passes.
//   yolo_sumsq_f32 : accumulates sum(g^2) of one tensor into a device double (fp64 atomics)
//   yolo_adam_step : g' = g * clip + wd * p ; m,v update ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
//                    clip = min(1, max_norm / (sqrt(norm_sq) + 1e-6)) is read from device memory, so the
//                    whole step needs no host synchronisation.  Optionally also writes bf16(p) (the
//                    forward operand of Linear layers has the master's layout).
// 16 B per lane on every stream (float4), grid-stride, <= 2048 workgroups.
//
// Multi-tensor forms (yolo_sumsq_f32_multi, yolo_adam_step_multi): the model has 52 parameter tensors,
// 48 of them small; one launch per tensor leaves most of the chip idle for most of the step.  The
// per-tensor table travels in the kernel arguments (<= YOLO_MT_MAX entries per launch), a workgroup
// owns one MT_CHUNK-element slice of one tensor and finds it by scanning the table's chunk prefix.
#include "common.h"

namespace yolo {

__global__ void __launch_bounds__(256) sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ acc)
{
    const long stride = (long)gridDim.x * blockDim.x * 4;
    double s = 0.0;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4 *>(g + i);
            s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
        } else {
            for (long k = i; k < n; ++k) s += (double)(g[k] * g[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float clip, float wd, float b1, float b2, float step_size, float inv_bc2_sqrt, float eps)
{
    g = g * clip;
    g = g + wd * p;                       // grad.add(param, alpha=weight_decay)
    m = m + (g - m) * (1.0f - b1);        // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (1.0f - b2) * g * g;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);      // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                                                   float b1, float b2, float eps, float wd, float step_size, float inv_bc2_sqrt,
                                                   const double *__restrict__ norm_sq, float max_norm, bf16_t *__restrict__ pb)
{
    float clip = 1.0f;
    if (norm_sq) {
        const float total = (float)sqrt(*norm_sq);
        const float c = max_norm / (total + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pv = *reinterpret_cast<float4 *>(p + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            float4 mv = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            adam1(pv.x, gv.x, mv.x, vv.x, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.y, gv.y, mv.y, vv.y, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.z, gv.z, mv.z, vv.z, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.w, gv.w, mv.w, vv.w, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            *reinterpret_cast<float4 *>(p + i) = pv;
            *reinterpret_cast<float4 *>(m + i) = mv;
            *reinterpret_cast<float4 *>(v + i) = vv;
            if (pb) {
                uint2 o;
                o.x = (unsigned)f32_to_bf16(pv.x) | ((unsigned)f32_to_bf16(pv.y) << 16);
                o.y = (unsigned)f32_to_bf16(pv.z) | ((unsigned)f32_to_bf16(pv.w) << 16);
                *reinterpret_cast<uint2 *>(pb + i) = o;
            }
        } else {
            for (long k = i; k < n; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam1(pk, g[k], mk, vk, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
                if (pb) pb[k] = f32_to_bf16(pk);
            }
        }
    }
}

constexpr int MT_CHUNK = 8192;  // elements per workgroup: 256 lanes x float4 x 8
constexpr int SQ_CHUNK = 65536; // the norm kernel ends in ONE fp64 atomic per workgroup on one address: keep them few

struct SumsqTable {
    const float *g[YOLO_MT_MAX];
    long n[YOLO_MT_MAX];
    int first[YOLO_MT_MAX + 1];   // first chunk (= workgroup) of every tensor
    int count;
};
struct AdamTable {
    yolo_adam_tensor t[YOLO_MT_MAX];
    int first[YOLO_MT_MAX + 1];
    int count;
};

__device__ __forceinline__ int find_tensor(const int *first, int count, int b)
{
    int i = 0;
    while (i + 1 < count && first[i + 1] <= b) ++i;  // wave-uniform scalar scan of <= 48 entries
    return i;
}

__global__ void __launch_bounds__(256) sumsq_multi_kernel(const SumsqTable tab, double *__restrict__ acc)
{
    const int ti = find_tensor(tab.first, tab.count, blockIdx.x);
    const float *__restrict__ g = tab.g[ti];
    const long n = tab.n[ti];
    const long beg = (long)(blockIdx.x - tab.first[ti]) * SQ_CHUNK;
    const long end = min(n, beg + SQ_CHUNK);
    double s = 0.0;
    for (long i = beg + threadIdx.x * 4; i < end; i += 1024) {
        if (i + 4 <= end) {
            const float4 v = *reinterpret_cast<const float4 *>(g + i);
            s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
        } else {
            for (long k = i; k < end; ++k) s += (double)(g[k] * g[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamTable tab, float b1, float b2, float eps, float wd, float step_size, float inv_bc2_sqrt,
                                                         const double *__restrict__ norm_sq, float max_norm, const float *__restrict__ skip_flag)
{
    if (skip_flag && *skip_flag != 0.0f) return;      // the producer of the gradients flagged this step as invalid: nothing is updated
    float clip = 1.0f;
    if (norm_sq) {
        const float total = (float)sqrt(*norm_sq);
        const float c = max_norm / (total + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    const int ti = find_tensor(tab.first, tab.count, blockIdx.x);
    float *__restrict__ p = tab.t[ti].p;
    const float *__restrict__ g = tab.t[ti].g;
    float *__restrict__ m = tab.t[ti].m;
    float *__restrict__ v = tab.t[ti].v;
    bf16_t *__restrict__ pb = (bf16_t *)tab.t[ti].p_bf16;
    const long n = tab.t[ti].n;
    const long beg = (long)(blockIdx.x - tab.first[ti]) * MT_CHUNK;
    const long end = min(n, beg + MT_CHUNK);
    for (long i = beg + threadIdx.x * 4; i < end; i += 1024) {
        if (i + 4 <= end) {
            float4 pv = *reinterpret_cast<float4 *>(p + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            float4 mv = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            adam1(pv.x, gv.x, mv.x, vv.x, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.y, gv.y, mv.y, vv.y, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.z, gv.z, mv.z, vv.z, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            adam1(pv.w, gv.w, mv.w, vv.w, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
            *reinterpret_cast<float4 *>(p + i) = pv;
            *reinterpret_cast<float4 *>(m + i) = mv;
            *reinterpret_cast<float4 *>(v + i) = vv;
            if (pb) {
                uint2 o;
                o.x = (unsigned)f32_to_bf16(pv.x) | ((unsigned)f32_to_bf16(pv.y) << 16);
                o.y = (unsigned)f32_to_bf16(pv.z) | ((unsigned)f32_to_bf16(pv.w) << 16);
                *reinterpret_cast<uint2 *>(pb + i) = o;
            }
        } else {
            for (long k = i; k < end; ++k) {
                float pk = p[k], mk = m[k], vk = v[k];
                adam1(pk, g[k], mk, vk, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
                if (pb) pb[k] = f32_to_bf16(pk);
            }
        }
    }
}

// Background form (yolo_adam_step_multi_bg): `gridDim.x` PERSISTENT workgroups of 1024 threads walk the chunk list; the dynamic LDS
// they reserve (unused) keeps every other workgroup off their CU.  The pass then occupies exactly gridDim.x CUs -- HBM-bound work that
// runs beside the next forward's MFMA-bound conv stack on the remaining CUs instead of in front of it (a grid of 25 k small
// workgroups would starve, or be starved by, the conv kernels, whose workgroups need a whole CU each).  Four float4 per array and
// thread are in flight: ~190 KB per CU, what ~100 GB/s per CU needs at HBM latency.
__global__ void __launch_bounds__(1024) adam_multi_bg_kernel(const AdamTable tab, int chunks, float b1, float b2, float eps, float wd, float step_size,
                                                             float inv_bc2_sqrt, const double *__restrict__ norm_sq, float max_norm,
                                                             const float *__restrict__ skip_flag)
{
    if (skip_flag && *skip_flag != 0.0f) return;
    float clip = 1.0f;
    if (norm_sq) {
        const float total = (float)sqrt(*norm_sq);
        const float c = max_norm / (total + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    // chunk = MT_CHUNK elements = 1024 threads x 2 x float4; the loads of the NEXT chunk are issued before the current one is computed
    // and stored, so that a CU always has ~128-256 KB in flight (without the prefetch a pass on 48 CUs reached 44 GB/s per CU)
    struct Vals {
        float4 p[2], g[2], m[2], v[2];
    };
    auto where = [&](int b, int &ti, long &beg, bool &full) {
        ti = find_tensor(tab.first, tab.count, b);
        beg = (long)(b - tab.first[ti]) * MT_CHUNK;
        full = beg + MT_CHUNK <= tab.t[ti].n;
    };
    auto load = [&](int ti, long beg, Vals &x) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long i = beg + (long)(u * 1024 + threadIdx.x) * 4;
            x.p[u] = *reinterpret_cast<const float4 *>(tab.t[ti].p + i);
            x.g[u] = *reinterpret_cast<const float4 *>(tab.t[ti].g + i);
            x.m[u] = *reinterpret_cast<const float4 *>(tab.t[ti].m + i);
            x.v[u] = *reinterpret_cast<const float4 *>(tab.t[ti].v + i);
        }
    };
    int b = blockIdx.x;
    int ti = 0, nti = 0;
    long beg = 0, nbeg = 0;
    bool full = false, nfull = false;
    Vals cur, nxt;
    if (b < chunks) {
        where(b, ti, beg, full);
        if (full) load(ti, beg, cur);
    }
    while (b < chunks) {
        const int nb = b + (int)gridDim.x;
        if (nb < chunks) {
            where(nb, nti, nbeg, nfull);
            if (nfull) load(nti, nbeg, nxt);
        }
        float *__restrict__ p = tab.t[ti].p;
        float *__restrict__ m = tab.t[ti].m;
        float *__restrict__ v = tab.t[ti].v;
        bf16_t *__restrict__ pb = (bf16_t *)tab.t[ti].p_bf16;
        if (full) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long i = beg + (long)(u * 1024 + threadIdx.x) * 4;
                adam1(cur.p[u].x, cur.g[u].x, cur.m[u].x, cur.v[u].x, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                adam1(cur.p[u].y, cur.g[u].y, cur.m[u].y, cur.v[u].y, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                adam1(cur.p[u].z, cur.g[u].z, cur.m[u].z, cur.v[u].z, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                adam1(cur.p[u].w, cur.g[u].w, cur.m[u].w, cur.v[u].w, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                *reinterpret_cast<float4 *>(p + i) = cur.p[u];
                *reinterpret_cast<float4 *>(m + i) = cur.m[u];
                *reinterpret_cast<float4 *>(v + i) = cur.v[u];
                if (pb) {
                    uint2 o;
                    o.x = (unsigned)f32_to_bf16(cur.p[u].x) | ((unsigned)f32_to_bf16(cur.p[u].y) << 16);
                    o.y = (unsigned)f32_to_bf16(cur.p[u].z) | ((unsigned)f32_to_bf16(cur.p[u].w) << 16);
                    *reinterpret_cast<uint2 *>(pb + i) = o;
                }
            }
        } else {
            const float *__restrict__ g = tab.t[ti].g;
            const long end = min(tab.t[ti].n, beg + MT_CHUNK);
            for (long k = beg + threadIdx.x; k < end; k += 1024) {     // last, partial chunk of a tensor
                float pk = p[k], mk = m[k], vk = v[k];
                adam1(pk, g[k], mk, vk, clip, wd, b1, b2, step_size, inv_bc2_sqrt, eps);
                p[k] = pk; m[k] = mk; v[k] = vk;
                if (pb) pb[k] = f32_to_bf16(pk);
            }
        }
        b = nb; ti = nti; beg = nbeg; full = nfull;
        cur = nxt;
    }
}

__global__ void scale_by_clip_kernel(float *__restrict__ g, long n, const double *__restrict__ norm_sq, float max_norm)
{
    const float total = (float)sqrt(*norm_sq);
    const float c = max_norm / (total + 1e-6f);
    if (c >= 1.0f) return;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= c;
}

}  // namespace yolo

using namespace yolo;

static inline unsigned grid_for(long n, int per_thread)
{
    long b = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

YOLO_API int yolo_sumsq_f32(const float *g, long n, double *acc, yolo_stream_t stream)
{
    if (!g || !acc || n < 0) return fail(YOLO_E_ARG, "yolo_sumsq_f32: bad argument");
    if (n == 0) return 0;
    if ((uintptr_t)g & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_sumsq_f32: pointer must be 16-B aligned");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 16)), dim3(256), 0, STRM(stream), g, n, acc);
    return check_launch("yolo_sumsq_f32");
}

YOLO_API int yolo_adam_step(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                            long step, const double *norm_sq, float max_norm, void *p_bf16, yolo_stream_t stream)
{
    if (!p || !g || !m || !v || n < 0 || step < 1) return fail(YOLO_E_ARG, "yolo_adam_step: bad argument");
    if (n == 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step: pointers must be 16-B aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8)), dim3(256), 0, STRM(stream), p, g, m, v, n, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
                       norm_sq, max_norm, (bf16_t *)p_bf16);
    return check_launch("yolo_adam_step");
}

YOLO_API int yolo_sumsq_f32_multi(const float *const *g, const long *n, int count, double *acc, yolo_stream_t stream)
{
    if (!g || !n || !acc || count < 0) return fail(YOLO_E_ARG, "yolo_sumsq_f32_multi: bad argument");
    for (int base = 0; base < count;) {
        SumsqTable tab{};
        long chunks = 0;
        int k = 0;
        for (; base + k < count && k < YOLO_MT_MAX; ++k) {
            const float *gp = g[base + k];
            const long nn = n[base + k];
            if (!gp || nn < 0) return fail(YOLO_E_ARG, "yolo_sumsq_f32_multi: tensor %d: null pointer or negative size", base + k);
            if ((uintptr_t)gp & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_sumsq_f32_multi: tensor %d is not 16-B aligned", base + k);
            const long c = (nn + SQ_CHUNK - 1) / SQ_CHUNK;
            if (chunks + c > 0x7fffffffL) break;
            tab.g[k] = gp; tab.n[k] = nn; tab.first[k] = (int)chunks;
            chunks += c;
        }
        if (k == 0) return fail(YOLO_E_UNSUPPORTED, "yolo_sumsq_f32_multi: tensor too large");
        tab.first[k] = (int)chunks;
        tab.count = k;
        if (chunks > 0) {
            hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, STRM(stream), tab, acc);
            if (int rc = check_launch("yolo_sumsq_f32_multi")) return rc;
        }
        base += k;
    }
    return 0;
}

YOLO_API int yolo_adam_step_multi(const yolo_adam_tensor *t, int count, float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                                  const double *norm_sq, float max_norm, const float *skip_flag, yolo_stream_t stream)
{
    if (!t || count < 0 || step < 1) return fail(YOLO_E_ARG, "yolo_adam_step_multi: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    for (int base = 0; base < count;) {
        AdamTable tab{};
        long chunks = 0;
        int k = 0;
        for (; base + k < count && k < YOLO_MT_MAX; ++k) {
            const yolo_adam_tensor &e = t[base + k];
            if (!e.p || !e.g || !e.m || !e.v || e.n < 0) return fail(YOLO_E_ARG, "yolo_adam_step_multi: tensor %d: null pointer or negative size", base + k);
            if (((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi: tensor %d is not 16-B aligned", base + k);
            if ((uintptr_t)e.p_bf16 & 7) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi: bf16 shadow %d is not 8-B aligned", base + k);
            const long c = (e.n + MT_CHUNK - 1) / MT_CHUNK;
            if (chunks + c > 0x7fffffffL) break;
            tab.t[k] = e; tab.first[k] = (int)chunks;
            chunks += c;
        }
        if (k == 0) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi: tensor too large");
        tab.first[k] = (int)chunks;
        tab.count = k;
        if (chunks > 0) {
            hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, STRM(stream), tab, beta1, beta2, eps, weight_decay, step_size,
                               inv_bc2_sqrt, norm_sq, max_norm, skip_flag);
            if (int rc = check_launch("yolo_adam_step_multi")) return rc;
        }
        base += k;
    }
    return 0;
}

YOLO_API int yolo_adam_step_multi_bg(const yolo_adam_tensor *t, int count, float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                                     const double *norm_sq, float max_norm, const float *skip_flag, int workgroups, yolo_stream_t stream)
{
    if (!t || count < 0 || count > YOLO_MT_MAX || step < 1 || workgroups < 1 || workgroups > 256)
        return fail(YOLO_E_ARG, "yolo_adam_step_multi_bg: bad argument (at most %d tensors, 1 .. 256 workgroups)", YOLO_MT_MAX);
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    AdamTable tab{};
    long chunks = 0;
    for (int k = 0; k < count; ++k) {
        const yolo_adam_tensor &e = t[k];
        if (!e.p || !e.g || !e.m || !e.v || e.n < 0) return fail(YOLO_E_ARG, "yolo_adam_step_multi_bg: tensor %d: null pointer or negative size", k);
        if (((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi_bg: tensor %d is not 16-B aligned", k);
        if ((uintptr_t)e.p_bf16 & 7) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi_bg: bf16 shadow %d is not 8-B aligned", k);
        tab.t[k] = e; tab.first[k] = (int)chunks;
        chunks += (e.n + MT_CHUNK - 1) / MT_CHUNK;
        if (chunks > 0x7fffffffL) return fail(YOLO_E_UNSUPPORTED, "yolo_adam_step_multi_bg: too many elements");
    }
    tab.first[count] = (int)chunks;
    tab.count = count;
    if (chunks == 0) return 0;
    constexpr int BG_LDS = 96 * 1024;       // with 1024 threads: one such workgroup per CU, and no 128-KB conv workgroup beside it
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)adam_multi_bg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BG_LDS);
        if (e != hipSuccess) return fail((int)e, "yolo_adam_step_multi_bg: hipFuncSetAttribute(%d B LDS): %s", BG_LDS, hipGetErrorString(e));
        attr_done[dev] = true;
    }
    const int G = (int)std::min<long>(workgroups, chunks);
    hipLaunchKernelGGL(adam_multi_bg_kernel, dim3((unsigned)G), dim3(1024), BG_LDS, STRM(stream), tab, (int)chunks, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt,
                       norm_sq, max_norm, skip_flag);
    return check_launch("yolo_adam_step_multi_bg");
}

YOLO_API int yolo_clip_scale_f32(float *g, long n, const double *norm_sq, float max_norm, yolo_stream_t stream)
{
    if (!g || !norm_sq || n < 0) return fail(YOLO_E_ARG, "yolo_clip_scale_f32: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_by_clip_kernel, dim3(grid_for(n, 4)), dim3(256), 0, STRM(stream), g, n, norm_sq, max_norm);
    return check_launch("yolo_clip_scale_f32");
}
