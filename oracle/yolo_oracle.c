/*
 * oracle/yolo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar CPU restatement of the reference's hot path (mattiaskvist/yolo-v1), used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker beside the
 * HIP kernels.  Nothing under yolo-v1_amd/ may link, import or call this file.
 *
 * Parity status: PINNED -- every function below is checked against fixtures produced by running
 * the reference itself (tests/golden/make_golden.py -> tests/golden/ npz files) and against the
 * known-answer tests the reference's own test-suite holds (tests/test_yolo.py:196-313,
 * tests/test_metrics.py:35-117,208-222 of the reference), see tests/test_oracle_*.py.
 *
 * Arithmetic conventions (must not be "optimised"):
 *   - decode / IoU / NMS follow the reference's Python-float maths: fp32 tensor elements are
 *     widened to double by .item() and every later operation is an IEEE double operation in
 *     the reference's source order.  Build with -ffp-contract=off (no FMA fusion).
 *   - the loss follows torch's fp32 elementwise order per cell; the five sums are accumulated
 *     in double and rounded once (torch.sum uses an fp32 tree; the fixtures agree to <=1e-6 rel).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * YOLO loss, forward + analytic backward.
 * Follows src/yolo/loss.py:87-172 (forward) and :174-212 (compute_iou); the backward is the
 * hand derivation of what autograd does for those lines, including
 *   - clamp(min=c) passing gradient where input >= c      (torch clamp_backward, inclusive)
 *   - maximum/minimum splitting the gradient 1/2 on ties  (torch derivatives.yaml)
 *   - best_ious NOT detached: d(conf_obj)/d(x,y,w,h) flows through the IoU (loss.py:111,123,144)
 * Returns 0, or -1 if some object cell selects a target slot >= B (the reference's gather
 * raises "index out of bounds" there: SURVEY.md 8a step 2).
 * out5 = {total, coord, conf_obj, conf_noobj, class}, each already divided by N (loss.py:162-170).
 * dpred (may be NULL) = d total / d pred.
 * ------------------------------------------------------------------------------------------ */
static float maxgrad(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); } /* d max(a,b)/da */
static float mingrad(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); } /* d min(a,b)/da */

typedef struct { float iou, inter, U, iw, ih, gx1, gx2, gy1, gy2, dw_pass, dh_pass; } iou_ctx;

static void iou_fwd(const float *p, const float *t, iou_ctx *c)
{
    /* loss.py:191-212, fp32, same operation order */
    float x1 = p[0] - p[2] / 2, y1 = p[1] - p[3] / 2, x2 = p[0] + p[2] / 2, y2 = p[1] + p[3] / 2;
    float tx1 = t[0] - t[2] / 2, ty1 = t[1] - t[3] / 2, tx2 = t[0] + t[2] / 2, ty2 = t[1] + t[3] / 2;
    float ix1 = x1 > tx1 ? x1 : tx1, iy1 = y1 > ty1 ? y1 : ty1;
    float ix2 = x2 < tx2 ? x2 : tx2, iy2 = y2 < ty2 ? y2 : ty2;
    float dw = ix2 - ix1, dh = iy2 - iy1;
    c->iw = dw < 0 ? 0.0f : dw;
    c->ih = dh < 0 ? 0.0f : dh;
    c->dw_pass = dw >= 0 ? 1.0f : 0.0f;
    c->dh_pass = dh >= 0 ? 1.0f : 0.0f;
    c->inter = c->iw * c->ih;
    float a1 = p[2] * p[3], a2 = t[2] * t[3];
    float uni = a1 + a2 - c->inter;
    c->U = uni + 1e-6f;
    c->iou = c->inter / c->U;
    c->gx1 = maxgrad(x1, tx1);
    c->gy1 = maxgrad(y1, ty1);
    c->gx2 = mingrad(x2, tx2);
    c->gy2 = mingrad(y2, ty2);
}

ORACLE_API int oracle_loss_fwd_bwd(const float *pred, const float *tgt, int N, int S, int B, int C,
                                   float lambda_coord, float lambda_noobj, float *out5, float *dpred)
{
    const int D = B * 5 + C;
    const long cells = (long)N * S * S;
    double s_coord = 0, s_obj = 0, s_noobj = 0, s_cls = 0;
    const float invN = 1.0f / (float)N;
    int err = 0;
    if (dpred) memset(dpred, 0, sizeof(float) * cells * D);
    for (long cell = 0; cell < cells; ++cell) {
        const float *p = pred + cell * D, *t = tgt + cell * D;
        float *g = dpred ? dpred + cell * D : NULL;
        /* loss.py:98-102: mask over channels 4,9,14,... of ALL D channels; idx = first set one */
        int obj = 0, idx = 0;
        for (int k = 0, ch = 4; ch < D; ch += 5, ++k)
            if (t[ch] > 0) { if (!obj) idx = k; obj = 1; }
        if (!obj) {
            for (int b = 0; b < B; ++b) {
                float c = p[b * 5 + 4];
                s_noobj += (double)(c * c);
                if (g) g[b * 5 + 4] = lambda_noobj * 2.0f * c * invN;
            }
            continue;
        }
        if (idx >= B) { err = -1; continue; }
        const float *tb = t + idx * 5;
        /* loss.py:107-111 */
        iou_ctx ctx[16];
        int best = 0;
        for (int b = 0; b < B && b < 16; ++b) {
            iou_fwd(p + b * 5, tb, &ctx[b]);
            if (ctx[b].iou > ctx[best].iou) best = b; /* argmax: first maximum */
        }
        const float *pr = p + best * 5;
        const iou_ctx *c = &ctx[best];
        /* loss.py:127-137 */
        float dx = pr[0] - tb[0], dy = pr[1] - tb[1];
        float cw = pr[2] < 1e-6f ? 1e-6f : pr[2], chh = pr[3] < 1e-6f ? 1e-6f : pr[3];
        float ctw = tb[2] < 1e-6f ? 1e-6f : tb[2], cth = tb[3] < 1e-6f ? 1e-6f : tb[3];
        float sw = sqrtf(cw), sh = sqrtf(chh);
        float ew = sw - sqrtf(ctw), eh = sh - sqrtf(cth);
        s_coord += (double)(dx * dx) + (double)(dy * dy) + (double)(ew * ew) + (double)(eh * eh);
        /* loss.py:142-144 */
        float ec = pr[4] - c->iou;
        s_obj += (double)(ec * ec);
        /* loss.py:150-153: every non-responsible box, including the loser of this cell */
        for (int b = 0; b < B; ++b)
            if (b != best) { float cc = p[b * 5 + 4]; s_noobj += (double)(cc * cc); if (g) g[b * 5 + 4] = lambda_noobj * 2.0f * cc * invN; }
        /* loss.py:156-157 */
        for (int k = 0; k < C; ++k) {
            float e = p[B * 5 + k] - t[B * 5 + k];
            s_cls += (double)(e * e);
            if (g) g[B * 5 + k] = 2.0f * e * invN;
        }
        if (g) {
            float *gr = g + best * 5;
            float g_iou = -2.0f * ec;                       /* d conf_obj / d iou */
            float inv_u = 1.0f / c->U;
            float d_inter = g_iou * (inv_u + c->inter * inv_u * inv_u); /* union = a1+a2-inter */
            float d_area = -g_iou * c->inter * inv_u * inv_u;
            float d_iw = d_inter * c->ih * c->dw_pass;      /* through clamp(min=0) */
            float d_ih = d_inter * c->iw * c->dh_pass;
            /* iw = ix2 - ix1 ; ix1 = max(x1,tx1) ; ix2 = min(x2,tx2) */
            float d_x1 = -d_iw * c->gx1, d_x2 = d_iw * c->gx2;
            float d_y1 = -d_ih * c->gy1, d_y2 = d_ih * c->gy2;
            float gx = d_x1 + d_x2, gy = d_y1 + d_y2;
            float gw = 0.5f * (d_x2 - d_x1) + d_area * pr[3];
            float gh = 0.5f * (d_y2 - d_y1) + d_area * pr[2];
            gx += lambda_coord * 2.0f * dx;
            gy += lambda_coord * 2.0f * dy;
            if (pr[2] >= 1e-6f) gw += lambda_coord * 2.0f * ew * (0.5f / sw);
            if (pr[3] >= 1e-6f) gh += lambda_coord * 2.0f * eh * (0.5f / sh);
            gr[0] = gx * invN; gr[1] = gy * invN; gr[2] = gw * invN; gr[3] = gh * invN;
            gr[4] = 2.0f * ec * invN;
        }
    }
    float coord = (float)((double)lambda_coord * s_coord), cobj = (float)s_obj;
    float cno = (float)((double)lambda_noobj * s_noobj), cls = (float)s_cls;
    out5[0] = (coord + cobj + cno + cls) / (float)N;
    out5[1] = coord / (float)N; out5[2] = cobj / (float)N; out5[3] = cno / (float)N; out5[4] = cls / (float)N;
    return err;
}

/* loss.py:174-212 as a free function over n box pairs (x,y,w,h), fp32 */
ORACLE_API void oracle_loss_iou(const float *b1, const float *b2, long n, float *out)
{
    iou_ctx c;
    for (long i = 0; i < n; ++i) { iou_fwd(b1 + 4 * i, b2 + 4 * i, &c); out[i] = c.iou; }
}

/* ------------------------------------------------------------------------------------------
 * Decode.  rec[k] = {class_id, conf, x, y, w, h} as doubles, scan order (i, j, b).
 * Predictions: src/yolo/inference.py:170-210 == src/yolo/metrics.py:185-218.
 * Ground truth: src/yolo/metrics.py:232-256 (slot 0 only, rec = {class_id, x, y, w, h}).
 * ------------------------------------------------------------------------------------------ */
static int argmax_first(const float *v, int n)
{
    int a = 0;
    for (int k = 1; k < n; ++k) if (v[k] > v[a]) a = k;
    return a;
}

ORACLE_API int oracle_decode(const float *pred, int S, int B, int C, double conf_thr, double *rec)
{
    const int D = B * 5 + C;
    int n = 0;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const float *cell = pred + ((long)i * S + j) * D;
            int cls = argmax_first(cell + B * 5, C);
            double prob = (double)cell[B * 5 + cls];
            for (int b = 0; b < B; ++b) {
                const float *bx = cell + b * 5;
                double x = ((double)j + (double)bx[0]) / (double)S;
                double y = ((double)i + (double)bx[1]) / (double)S;
                double fc = (double)bx[4] * prob;
                if (fc > conf_thr) {
                    double *r = rec + 6 * (long)n++;
                    r[0] = cls; r[1] = fc; r[2] = x; r[3] = y; r[4] = (double)bx[2]; r[5] = (double)bx[3];
                }
            }
        }
    return n;
}

ORACLE_API int oracle_decode_gt(const float *tgt, int S, int B, int C, double *rec)
{
    const int D = B * 5 + C;
    int n = 0;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            const float *cell = tgt + ((long)i * S + j) * D;
            if (cell[4] > 0) {
                double *r = rec + 5 * (long)n++;
                r[0] = argmax_first(cell + B * 5, C);
                r[1] = ((double)j + (double)cell[0]) / (double)S;
                r[2] = ((double)i + (double)cell[1]) / (double)S;
                r[3] = (double)cell[2]; r[4] = (double)cell[3];
            }
        }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Scalar IoU, two formulas.  box = (x, y, w, h) doubles.
 * variant 0 "inference": schemas.py:18-29,52-55 + inference.py:229-249  (+1e-6 in the denominator)
 * variant 1 "metrics":   metrics.py:313-341                              (union==0 -> 0.0, no eps)
 * Python's max(0, v) returns 0 unless v > 0.
 * ------------------------------------------------------------------------------------------ */
ORACLE_API double oracle_iou(const double *a, const double *b, int variant)
{
    double ax1 = a[0] - a[2] / 2, ay1 = a[1] - a[3] / 2, ax2 = a[0] + a[2] / 2, ay2 = a[1] + a[3] / 2;
    double bx1 = b[0] - b[2] / 2, by1 = b[1] - b[3] / 2, bx2 = b[0] + b[2] / 2, by2 = b[1] + b[3] / 2;
    double ix1 = bx1 > ax1 ? bx1 : ax1;   /* max(x1_min, x2_min): returns the 2nd only if greater */
    double iy1 = by1 > ay1 ? by1 : ay1;
    double ix2 = bx2 < ax2 ? bx2 : ax2;
    double iy2 = by2 < ay2 ? by2 : ay2;
    double dw = ix2 - ix1, dh = iy2 - iy1;
    double inter = (dw > 0 ? dw : 0.0) * (dh > 0 ? dh : 0.0);
    double a1 = a[2] * a[3], a2 = b[2] * b[3];
    if (variant == 0) return inter / (a1 + a2 - inter + 1e-6);
    double uni = a1 + a2 - inter;
    if (uni == 0) return 0.0;
    return inter / uni;
}

/* ------------------------------------------------------------------------------------------
 * NMS over n decoded records (6 doubles each).  keep[] receives indices into rec in OUTPUT order.
 * variant 0: inference.py:298-317 -- one list, stable sort by conf desc, greedy, a later box
 *            survives a kept one iff class differs or IoU(eps) < thr; output in confidence order.
 * variant 1: metrics.py:270-296 -- same sort, then per-class lists in first-appearance order
 *            (defaultdict insertion order), greedy with IoU(no eps) < thr; output grouped by class.
 * ------------------------------------------------------------------------------------------ */
ORACLE_API int oracle_nms(const double *rec, int n, double thr, int variant, int *keep)
{
    if (n <= 0) return 0;
    int *order = (int *)malloc(sizeof(int) * n);
    char *dead = (char *)calloc(n, 1);
    for (int i = 0; i < n; ++i) order[i] = i;
    /* stable insertion sort, descending conf: sorted(..., reverse=True) keeps ties in input order */
    for (int i = 1; i < n; ++i) {
        int v = order[i], j = i - 1;
        while (j >= 0 && rec[6 * order[j] + 1] < rec[6 * v + 1]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
    int nk = 0;
    if (variant == 0) {
        for (int a = 0; a < n; ++a) {
            if (dead[a]) continue;
            int ia = order[a];
            keep[nk++] = ia;
            for (int b = a + 1; b < n; ++b) {
                if (dead[b]) continue;
                int ib = order[b];
                if (rec[6 * ib] != rec[6 * ia]) continue;
                if (!(oracle_iou(rec + 6 * ia + 2, rec + 6 * ib + 2, 0) < thr)) dead[b] = 1;
            }
        }
    } else {
        char *done = (char *)calloc(n, 1);
        for (int f = 0; f < n; ++f) {
            if (done[f]) continue;
            double cls = rec[6 * order[f]];
            for (int a = f; a < n; ++a) {
                if (rec[6 * order[a]] != cls) continue;
                done[a] = 1;
                if (dead[a]) continue;
                int ia = order[a];
                keep[nk++] = ia;
                for (int b = a + 1; b < n; ++b) {
                    int ib = order[b];
                    if (dead[b] || rec[6 * ib] != cls) continue;
                    if (!(oracle_iou(rec + 6 * ia + 2, rec + 6 * ib + 2, 1) < thr)) dead[b] = 1;
                }
            }
        }
        free(done);
    }
    free(order);
    free(dead);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * Naive fp32 layers with the reference's hyper-parameters (src/yolo/models.py:47-84,239-245):
 * Conv2d(bias) NCHW/OIHW + optional LeakyReLU, MaxPool2d(2,2), Linear.  Double accumulation.
 * Only for tiny shapes (tests) -- the full-size network check uses stock torch.nn on the host,
 * which IS the reference's arithmetic for these layers (oracle/torch_ref.py).
 * ------------------------------------------------------------------------------------------ */
ORACLE_API void oracle_conv2d(const float *x, const float *w, const float *bias, float *y,
                              int N, int Ci, int H, int W, int Co, int K, int stride, int pad, float slope)
{
    int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int ky = 0; ky < K; ++ky) {
                            int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                int ix = ox * stride - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[(((long)n * Ci + ci) * H + iy) * W + ix] * (double)w[(((long)co * Ci + ci) * K + ky) * K + kx];
                            }
                        }
                    float v = (float)acc;
                    if (slope != 1.0f && v < 0) v *= slope;
                    y[(((long)n * Co + co) * Ho + oy) * Wo + ox] = v;
                }
}

ORACLE_API void oracle_maxpool2(const float *x, float *y, int N, int C, int H, int W)
{
    int Ho = H / 2, Wo = W / 2;
    for (long nc = 0; nc < (long)N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float *p = x + (nc * H + 2 * oy) * W + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                y[(nc * Ho + oy) * Wo + ox] = m;
            }
}

ORACLE_API void oracle_linear(const float *x, const float *w, const float *bias, float *y, int N, int K, int O, float slope)
{
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < O; ++o) {
            double acc = bias ? bias[o] : 0.0;
            for (int k = 0; k < K; ++k) acc += (double)x[(long)n * K + k] * (double)w[(long)o * K + k];
            float v = (float)acc;
            if (slope != 1.0f && v < 0) v *= slope;
            y[(long)n * O + o] = v;
        }
}
