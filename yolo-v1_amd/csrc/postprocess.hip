// S x S x B box decode, pairwise IoU and per-image NMS for gfx950 -- wavefront primitives
// (64-lane ballot / popcount prefix), no MFMA.  Everything is fp64 arithmetic on fp32 inputs in
// the reference's Python operation order; this file MUST be compiled with -ffp-contract=off so
// that no multiply-add pair is fused (bit-exact confidences / IoUs => bit-exact kept indices).
//
// Reference behaviour restated (mattiaskvist/yolo-v1):
//   decode   src/yolo/inference.py:170-210, src/yolo/metrics.py:185-218, :232-256
//   IoU      src/yolo/inference.py:229-249 + src/yolo/schemas.py:18-55 ; src/yolo/metrics.py:313-341
//   NMS      src/yolo/inference.py:298-317 ; src/yolo/metrics.py:270-296
#include "common.h"

namespace yolo {

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// decode: one workgroup (128 threads = 2 waves) per image; candidate t = (cell, box) in scan order.
// Compaction keeps scan order: rank inside a wave = popcount of the ballot below the lane, waves
// are chained through LDS.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) decode_kernel(const float *__restrict__ pred, int S, int B, int C, double thr,
                                                     double *__restrict__ rec, int *__restrict__ counts)
{
    const int D = B * 5 + C;
    const int ncand = S * S * B;
    const int img = blockIdx.x;
    const float *p = pred + (size_t)img * S * S * D;
    double *out = rec + (size_t)img * ncand * 6;
    __shared__ int wave_cnt[2];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ncand; t0 += 128) {
        const int t = t0 + threadIdx.x;
        bool keep = false;
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0;
        if (t < ncand) {
            const int cell = t / B, b = t - cell * B;
            const int i = cell / S, j = cell - i * S;
            const float *c = p + (size_t)cell * D;
            // torch.argmax: first maximum
            int cls = 0;
            float best = c[B * 5];
            for (int k = 1; k < C; ++k) {
                float v = c[B * 5 + k];
                if (v > best) { best = v; cls = k; }
            }
            const float *bx = c + b * 5;
            const double prob = (double)best;
            r2 = ((double)j + (double)bx[0]) / (double)S;
            r3 = ((double)i + (double)bx[1]) / (double)S;
            r4 = (double)bx[2];
            r5 = (double)bx[3];
            r1 = (double)bx[4] * prob;
            r0 = (double)cls;
            keep = r1 > thr;
        }
        const unsigned long long m = __ballot(keep);
        const int below = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        const int base = base_s + (wave == 1 ? wave_cnt[0] : 0);
        if (keep) {
            double *r = out + (size_t)(base + below) * 6;
            r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4; r[5] = r5;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[img] = base_s;
}

__global__ void __launch_bounds__(128) decode_gt_kernel(const float *__restrict__ tgt, int S, int B, int C,
                                                        double *__restrict__ rec, int *__restrict__ counts)
{
    const int D = B * 5 + C;
    const int ncell = S * S;
    const int img = blockIdx.x;
    const float *p = tgt + (size_t)img * ncell * D;
    double *out = rec + (size_t)img * ncell * 5;
    __shared__ int wave_cnt[2];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ncell; t0 += 128) {
        const int cell = t0 + threadIdx.x;
        bool keep = false;
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
        if (cell < ncell) {
            const int i = cell / S, j = cell - i * S;
            const float *c = p + (size_t)cell * D;
            keep = c[4] > 0.0f;  // slot 0 only (metrics.py:239)
            int cls = 0;
            float best = c[B * 5];
            for (int k = 1; k < C; ++k) {
                float v = c[B * 5 + k];
                if (v > best) { best = v; cls = k; }
            }
            r0 = (double)cls;
            r1 = ((double)j + (double)c[0]) / (double)S;
            r2 = ((double)i + (double)c[1]) / (double)S;
            r3 = (double)c[2];
            r4 = (double)c[3];
        }
        const unsigned long long m = __ballot(keep);
        const int below = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        const int base = base_s + (wave == 1 ? wave_cnt[0] : 0);
        if (keep) {
            double *r = out + (size_t)(base + below) * 5;
            r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[img] = base_s;
}

// ------------------------------------------------------------------------------------------------
// scalar IoU, both reference formulas.  Python's max(a, b) returns a unless b > a; max(0, v) is 0
// unless v > 0.
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__device__ __forceinline__ double iou_f64(double ax, double ay, double aw, double ah, double bx, double by, double bw, double bh)
{
    const double ax1 = ax - aw / 2, ay1 = ay - ah / 2, ax2 = ax + aw / 2, ay2 = ay + ah / 2;
    const double bx1 = bx - bw / 2, by1 = by - bh / 2, bx2 = bx + bw / 2, by2 = by + bh / 2;
    const double ix1 = bx1 > ax1 ? bx1 : ax1;
    const double iy1 = by1 > ay1 ? by1 : ay1;
    const double ix2 = bx2 < ax2 ? bx2 : ax2;
    const double iy2 = by2 < ay2 ? by2 : ay2;
    const double dw = ix2 - ix1, dh = iy2 - iy1;
    const double inter = (dw > 0 ? dw : 0.0) * (dh > 0 ? dh : 0.0);
    const double a1 = aw * ah, a2 = bw * bh;
    if (VARIANT == YOLO_NMS_INFERENCE) return inter / (a1 + a2 - inter + 1e-6);
    const double uni = a1 + a2 - inter;
    if (uni == 0) return 0.0;
    return inter / uni;
}

template <int VARIANT>
__global__ void pairwise_iou_kernel(const double *__restrict__ a, int na, const double *__restrict__ b, int nb, double *__restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)na * nb) return;
    const int i = (int)(idx / nb), j = (int)(idx - (long)i * nb);
    const double *p = a + 4 * (size_t)i, *q = b + 4 * (size_t)j;
    out[idx] = iou_f64<VARIANT>(p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]);
}

// ------------------------------------------------------------------------------------------------
// NMS: one workgroup (4 wavefronts) per image, n <= 128 boxes.
//   1. stable descending rank by confidence (ties keep scan order, like sorted(reverse=True))
//   2. suppression matrix: row a (rank order) = the set of later same-class boxes b with
//      !(IoU(a,b) < thr), produced by two 64-lane ballots; the rows are dealt to the 4 waves, so
//      the fp64 IoU work (n^2/2 divisions) runs 256 lanes wide
//   3. greedy sweep: wave 0 walks the rows with two wave-uniform 64-bit "removed" masks -- 128
//      steps of bit arithmetic, no IoU, no barrier, no atomic
//   4. variant 1 re-orders the survivors by (first appearance of their class, rank)
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ void __launch_bounds__(256) nms_kernel(const double *__restrict__ rec, const int *__restrict__ counts, int max_per_img,
                                                  double thr, int *__restrict__ keep, int *__restrict__ keep_counts)
{
    const int img = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int n = counts[img];
    if (n > max_per_img) n = max_per_img;
    const double *r = rec + (size_t)img * max_per_img * 6;
    int *kout = keep + (size_t)img * max_per_img;

    __shared__ double s_conf[128];
    __shared__ double s_box[128][4];            // in rank order
    __shared__ double s_cls[128];               // in rank order
    __shared__ unsigned long long s_sup[128][2];  // suppression rows, rank order
    __shared__ int s_orig[128];                 // rank -> original index
    __shared__ int s_first[128];                // rank -> first rank holding the same class
    __shared__ int s_kept[128];                 // kept ranks, ascending
    __shared__ int s_nk;

    if (tid < n) s_conf[tid] = r[(size_t)tid * 6 + 1];
    __syncthreads();
    if (tid < n) {
        const double c = s_conf[tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double cj = s_conf[j];
            rank += (cj > c) || (cj == c && j < tid);
        }
        s_orig[rank] = tid;
        s_cls[rank] = r[(size_t)tid * 6 + 0];
        s_box[rank][0] = r[(size_t)tid * 6 + 2];
        s_box[rank][1] = r[(size_t)tid * 6 + 3];
        s_box[rank][2] = r[(size_t)tid * 6 + 4];
        s_box[rank][3] = r[(size_t)tid * 6 + 5];
    }
    __syncthreads();

    // this lane's two boxes (ranks lane and lane+64), the same in every wave
    const int b0 = lane, b1 = lane + 64;
    double c0 = -1, x0 = 0, y0 = 0, w0 = 0, h0 = 0, c1 = -1, x1 = 0, y1 = 0, w1 = 0, h1 = 0;
    if (b0 < n) { c0 = s_cls[b0]; x0 = s_box[b0][0]; y0 = s_box[b0][1]; w0 = s_box[b0][2]; h0 = s_box[b0][3]; }
    if (b1 < n) { c1 = s_cls[b1]; x1 = s_box[b1][0]; y1 = s_box[b1][1]; w1 = s_box[b1][2]; h1 = s_box[b1][3]; }
    for (int a = wave; a < n; a += 4) {
        const double ca = s_cls[a], xa = s_box[a][0], ya = s_box[a][1], wa = s_box[a][2], ha = s_box[a][3];
        bool p0 = false, p1 = false;
        if (b0 > a && b0 < n && c0 == ca) p0 = !(iou_f64<VARIANT>(xa, ya, wa, ha, x0, y0, w0, h0) < thr);
        if (b1 > a && b1 < n && c1 == ca) p1 = !(iou_f64<VARIANT>(xa, ya, wa, ha, x1, y1, w1, h1) < thr);
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        if (lane == 0) { s_sup[a][0] = m0; s_sup[a][1] = m1; }
    }
    if (VARIANT == YOLO_NMS_METRICS && tid < n) {
        const double ct = s_cls[tid];
        int first = tid;
        for (int j = 0; j < tid; ++j)
            if (s_cls[j] == ct) { first = j; break; }
        s_first[tid] = first;
    }
    __syncthreads();

    if (wave == 0) {
        unsigned long long removed0 = 0, removed1 = 0;  // wave-uniform
        int nk = 0;
        for (int a = 0; a < n; ++a) {
            const bool dead = a < 64 ? ((removed0 >> a) & 1ull) : ((removed1 >> (a - 64)) & 1ull);
            if (dead) continue;
            if (lane == 0) s_kept[nk] = a;
            ++nk;
            removed0 |= s_sup[a][0];
            removed1 |= s_sup[a][1];
        }
        if (lane == 0) s_nk = nk;
    }
    __syncthreads();
    const int nk = s_nk;

    if (VARIANT == YOLO_NMS_INFERENCE) {
        if (tid < nk) kout[tid] = s_orig[s_kept[tid]];
    } else if (tid < nk) {
        // class buckets in first-appearance order of the sorted list (dict insertion order):
        // position = number of survivors with a smaller (first, rank) key
        const int a = s_kept[tid], first = s_first[a];
        int pos = 0;
        for (int q = 0; q < nk; ++q) {
            const int aq = s_kept[q], fq = s_first[aq];
            pos += (fq < first) || (fq == first && aq < a);
        }
        kout[pos] = s_orig[a];
    }
    if (tid == 0) keep_counts[img] = nk;
}

// ------------------------------------------------------------------------------------------------
// NMS for 128 < n <= 1024 boxes per image (grids beyond S*S*B = 128, e.g. the reference's S = 14, B = 3 test model:
// 588 boxes).  Same order of comparisons and the same results as nms_kernel; the suppression matrix would not fit into LDS
// (1024 x 16 words), so the greedy sweep is done row by row: for every box that is still alive (rank order) the 256 threads
// test the later same-class boxes in parallel and OR their "removed" bits into a 1024-bit mask in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int NMS_BIG_MAX = 1024;

template <int VARIANT>
__global__ void __launch_bounds__(256) nms_big_kernel(const double *__restrict__ rec, const int *__restrict__ counts, int max_per_img,
                                                      double thr, int *__restrict__ keep, int *__restrict__ keep_counts)
{
    const int img = blockIdx.x;
    const int tid = threadIdx.x;
    int n = counts[img];
    if (n > max_per_img) n = max_per_img;
    const double *r = rec + (size_t)img * max_per_img * 6;
    int *kout = keep + (size_t)img * max_per_img;

    __shared__ double s_conf[NMS_BIG_MAX];         // scan order
    __shared__ double s_box[NMS_BIG_MAX][4];       // rank order
    __shared__ float s_cls[NMS_BIG_MAX];           // rank order (class ids are small integers: exact in fp32)
    __shared__ short s_orig[NMS_BIG_MAX], s_first[NMS_BIG_MAX], s_kept[NMS_BIG_MAX];
    __shared__ unsigned int s_removed[NMS_BIG_MAX / 32];
    __shared__ int s_nk;

    for (int i = tid; i < n; i += 256) s_conf[i] = r[(size_t)i * 6 + 1];
    if (tid < NMS_BIG_MAX / 32) s_removed[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const double c = s_conf[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double cj = s_conf[j];
            rank += (cj > c) || (cj == c && j < i);
        }
        s_orig[rank] = (short)i;
        s_cls[rank] = (float)r[(size_t)i * 6 + 0];
        s_box[rank][0] = r[(size_t)i * 6 + 2];
        s_box[rank][1] = r[(size_t)i * 6 + 3];
        s_box[rank][2] = r[(size_t)i * 6 + 4];
        s_box[rank][3] = r[(size_t)i * 6 + 5];
    }
    __syncthreads();
    if (VARIANT == YOLO_NMS_METRICS) {
        for (int i = tid; i < n; i += 256) {
            const float ct = s_cls[i];
            int first = i;
            for (int j = 0; j < i; ++j)
                if (s_cls[j] == ct) { first = j; break; }
            s_first[i] = (short)first;
        }
    }
    int nk = 0;                                    // the same value in every thread
    for (int a = 0; a < n; ++a) {
        if ((s_removed[a >> 5] >> (a & 31)) & 1u) continue;          // uniform: read after the barrier of the previous row
        if (tid == 0) s_kept[nk] = (short)a;
        ++nk;
        const float ca = s_cls[a];
        const double xa = s_box[a][0], ya = s_box[a][1], wa = s_box[a][2], ha = s_box[a][3];
        for (int b = a + 1 + tid; b < n; b += 256) {
            if (s_cls[b] != ca) continue;
            if (!(iou_f64<VARIANT>(xa, ya, wa, ha, s_box[b][0], s_box[b][1], s_box[b][2], s_box[b][3]) < thr)) atomicOr(&s_removed[b >> 5], 1u << (b & 31));
        }
        __syncthreads();
    }
    if (tid == 0) s_nk = nk;
    __syncthreads();
    if (VARIANT == YOLO_NMS_INFERENCE) {
        for (int i = tid; i < nk; i += 256) kout[i] = s_orig[s_kept[i]];
    } else {
        for (int i = tid; i < nk; i += 256) {
            const int a = s_kept[i], first = s_first[a];
            int pos = 0;
            for (int q = 0; q < nk; ++q) {
                const int aq = s_kept[q], fq = s_first[aq];
                pos += (fq < first) || (fq == first && aq < a);
            }
            kout[pos] = s_orig[a];
        }
    }
    if (tid == 0) keep_counts[img] = nk;
}

// ------------------------------------------------------------------------------------------------
// TP / FP matching of mAPMetric on the device (reference src/yolo/metrics.py:343-442, 568-651).
// The reference sorts ALL predictions of a class by confidence and walks them, but a prediction only ever meets
// the ground truths of its own image, and the kept list of an image (metrics NMS order: grouped by class,
// confidence-descending, ties in scan order) is exactly that global order restricted to the image -- so the whole
// greedy matching is per image.  One wavefront per image:
//   1. every kept prediction finds its best ground truth (largest IoU > 0, first on ties, same class) in four
//      ground-truth sets: all / small / medium / large (area limits of the reference's size metrics);
//   2. lane l = (set v, threshold t) walks the predictions in order with its own `taken` bitmask over the <= 64
//      ground truths: TP iff best_iou >= thr_t and that ground truth is still free.
// Output per kept prediction: bit v*(T+1)+t of tp_bits (t == T: the extra threshold of the overall
// precision / recall, set `all` only).  The host then only sorts, cumsums and interpolates (NumPy, vectorised).
// ------------------------------------------------------------------------------------------------
struct MapMatchParams {
    double thr[16];
    int T;
    double small_t, medium_t;
};

__global__ void __launch_bounds__(64) map_match_kernel(const double *__restrict__ rec, const int *__restrict__ keep, const int *__restrict__ kcnt, int M,
                                                       const double *__restrict__ grec, const int *__restrict__ gcnt, int G, const MapMatchParams prm,
                                                       unsigned long long *__restrict__ tp_bits, int *__restrict__ gt_bucket)
{
    __shared__ double g_box[64][4];
    __shared__ int g_cls[64], g_bkt[64];
    __shared__ double b_iou[4][128];
    __shared__ signed char b_gt[4][128];
    __shared__ unsigned char hit[128][64];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int nk = kcnt[img], ng = gcnt[img];
    if (lane < ng) {
        const double *g = grec + ((long)img * G + lane) * 5;
        g_cls[lane] = (int)g[0];
        g_box[lane][0] = g[1]; g_box[lane][1] = g[2]; g_box[lane][2] = g[3]; g_box[lane][3] = g[4];
        const double area = g[3] * g[4];
        const int b = area < prm.small_t ? 1 : (area < prm.medium_t ? 2 : 3);
        g_bkt[lane] = b;
        gt_bucket[(long)img * G + lane] = b;
    }
    __syncthreads();
    for (int k = lane; k < nk; k += 64) {
        const double *r = rec + ((long)img * M + keep[(long)img * M + k]) * 6;
        const int c = (int)r[0];
        double best[4] = {0.0, 0.0, 0.0, 0.0};
        int bg[4] = {-1, -1, -1, -1};
        for (int g = 0; g < ng; ++g) {
            if (g_cls[g] != c) continue;
            const double v = iou_f64<YOLO_NMS_METRICS>(r[2], r[3], r[4], r[5], g_box[g][0], g_box[g][1], g_box[g][2], g_box[g][3]);
            if (v > best[0]) { best[0] = v; bg[0] = g; }           // strict '>': first maximum, and IoU must be > 0
            const int b = g_bkt[g];
            if (v > best[b]) { best[b] = v; bg[b] = g; }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) { b_iou[v][k] = best[v]; b_gt[v][k] = (signed char)bg[v]; }
    }
    __syncthreads();
    const int T1 = prm.T + 1;
    if (lane < 4 * T1) {
        const int v = lane / T1, t = lane - v * T1;
        const double thr = prm.thr[t];
        unsigned long long taken = 0ull;
        for (int k = 0; k < nk; ++k) {
            const int g = b_gt[v][k];
            unsigned char h = 0;
            if (g >= 0 && b_iou[v][k] >= thr && !((taken >> g) & 1ull)) { taken |= 1ull << g; h = 1; }
            hit[k][lane] = h;
        }
    }
    __syncthreads();
    for (int k = lane; k < nk; k += 64) {
        unsigned long long bits = 0ull;
        for (int l = 0; l < 4 * T1; ++l) bits |= (unsigned long long)hit[k][l] << l;
        tp_bits[(long)img * M + k] = bits;
    }
}

}  // namespace yolo

using namespace yolo;

YOLO_API int yolo_decode(const float *pred, int N, int S, int B, int C, double conf_thr, double *rec, int32_t *counts, yolo_stream_t stream)
{
    if (!pred || !rec || !counts || N < 0 || S <= 0 || B <= 0 || C <= 0) return fail(YOLO_E_ARG, "yolo_decode: bad argument");
    if (S * S * B > 1024 || B > 8) return fail(YOLO_E_UNSUPPORTED, "yolo_decode: S*S*B=%d > 1024 or B=%d > 8", S * S * B, B);
    if (N == 0) return 0;
    hipLaunchKernelGGL(decode_kernel, dim3(N), dim3(128), 0, STRM(stream), pred, S, B, C, conf_thr, rec, counts);
    return check_launch("yolo_decode");
}

YOLO_API int yolo_decode_gt(const float *tgt, int N, int S, int B, int C, double *rec, int32_t *counts, yolo_stream_t stream)
{
    if (!tgt || !rec || !counts || N < 0 || S <= 0 || B <= 0 || C <= 0) return fail(YOLO_E_ARG, "yolo_decode_gt: bad argument");
    if (S * S > 1024) return fail(YOLO_E_UNSUPPORTED, "yolo_decode_gt: S*S=%d > 1024", S * S);
    if (N == 0) return 0;
    hipLaunchKernelGGL(decode_gt_kernel, dim3(N), dim3(128), 0, STRM(stream), tgt, S, B, C, rec, counts);
    return check_launch("yolo_decode_gt");
}

YOLO_API int yolo_nms(const double *rec, const int32_t *counts, int N, int max_per_img, double thr, int variant, int32_t *keep,
                      int32_t *keep_counts, yolo_stream_t stream)
{
    if (!rec || !counts || !keep || !keep_counts || N < 0 || max_per_img <= 0) return fail(YOLO_E_ARG, "yolo_nms: bad argument");
    if (max_per_img > NMS_BIG_MAX) return fail(YOLO_E_UNSUPPORTED, "yolo_nms: max_per_img=%d > %d", max_per_img, NMS_BIG_MAX);
    if (variant != YOLO_NMS_INFERENCE && variant != YOLO_NMS_METRICS) return fail(YOLO_E_ARG, "yolo_nms: variant %d", variant);
    if (N == 0) return 0;
    if (max_per_img > 128) {        // grids beyond S*S*B = 128: row-by-row sweep with a 1024-bit mask
        if (variant == YOLO_NMS_INFERENCE)
            hipLaunchKernelGGL(nms_big_kernel<YOLO_NMS_INFERENCE>, dim3(N), dim3(256), 0, STRM(stream), rec, counts, max_per_img, thr, keep, keep_counts);
        else
            hipLaunchKernelGGL(nms_big_kernel<YOLO_NMS_METRICS>, dim3(N), dim3(256), 0, STRM(stream), rec, counts, max_per_img, thr, keep, keep_counts);
        return check_launch("yolo_nms");
    }
    if (variant == YOLO_NMS_INFERENCE)
        hipLaunchKernelGGL(nms_kernel<YOLO_NMS_INFERENCE>, dim3(N), dim3(256), 0, STRM(stream), rec, counts, max_per_img, thr, keep, keep_counts);
    else
        hipLaunchKernelGGL(nms_kernel<YOLO_NMS_METRICS>, dim3(N), dim3(256), 0, STRM(stream), rec, counts, max_per_img, thr, keep, keep_counts);
    return check_launch("yolo_nms");
}

YOLO_API int yolo_pairwise_iou(const double *a, int na, const double *b, int nb, int variant, double *out, yolo_stream_t stream)
{
    if (!a || !b || !out || na < 0 || nb < 0) return fail(YOLO_E_ARG, "yolo_pairwise_iou: bad argument");
    if (variant != YOLO_NMS_INFERENCE && variant != YOLO_NMS_METRICS) return fail(YOLO_E_ARG, "yolo_pairwise_iou: variant %d", variant);
    const long total = (long)na * nb;
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256);
    if (variant == YOLO_NMS_INFERENCE)
        hipLaunchKernelGGL(pairwise_iou_kernel<YOLO_NMS_INFERENCE>, dim3(grid), dim3(256), 0, STRM(stream), a, na, b, nb, out);
    else
        hipLaunchKernelGGL(pairwise_iou_kernel<YOLO_NMS_METRICS>, dim3(grid), dim3(256), 0, STRM(stream), a, na, b, nb, out);
    return check_launch("yolo_pairwise_iou");
}

YOLO_API int yolo_map_match(const double *rec, const int32_t *keep, const int32_t *keep_counts, int N, int max_per_img, const double *gt_rec, const int32_t *gt_counts,
                            int max_gt, const double *thresholds, int T, double extra_threshold, double small_area, double medium_area,
                            unsigned long long *tp_bits, int32_t *gt_bucket, yolo_stream_t stream)
{
    if (!rec || !keep || !keep_counts || !gt_rec || !gt_counts || !thresholds || !tp_bits || !gt_bucket || N <= 0) return fail(YOLO_E_ARG, "yolo_map_match: bad argument");
    if (T < 1 || T > 15) return fail(YOLO_E_UNSUPPORTED, "yolo_map_match: 1..15 thresholds (got %d)", T);
    if (max_per_img < 1 || max_per_img > 128 || max_gt < 1 || max_gt > 64)
        return fail(YOLO_E_UNSUPPORTED, "yolo_map_match: at most 128 predictions and 64 ground truths per image (got %d, %d)", max_per_img, max_gt);
    MapMatchParams prm{};
    for (int t = 0; t < T; ++t) prm.thr[t] = thresholds[t];
    prm.thr[T] = extra_threshold;
    prm.T = T;
    prm.small_t = small_area; prm.medium_t = medium_area;
    hipLaunchKernelGGL(map_match_kernel, dim3(N), dim3(64), 0, STRM(stream), rec, keep, keep_counts, max_per_img, gt_rec, gt_counts, max_gt, prm, tp_bits, gt_bucket);
    return check_launch("yolo_map_match");
}
