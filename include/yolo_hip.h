/*
 * yolo_hip.h -- C ABI of libyolo_hip.so, the MI355X (gfx950) implementation of the YOLOv1 hot path.
 *
 * The reference (mattiaskvist/yolo-v1) is pure Python with no FFI layer; its hot path is the
 * arithmetic behind the `src/yolo` module surface.  This header is the boundary a
 * maintainer binds instead of the torch ops those modules call: every entry point names the
 * reference lines it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers
 *     unless a parameter is called host_*.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *     work; nothing here synchronises, allocates or frees device memory, so every call can be
 *     captured in a hipGraph.  Workspaces are caller-allocated.
 *   - return value: 0 = ok, <0 = YOLO_E_* argument error, >0 = hipError_t.  No exceptions cross
 *     the ABI.  yolo_hip_last_error() gives a human-readable string for the calling thread.
 *   - activations inside the network are "padded NHWC bf16": [N][H+2*halo][W+2*halo][C] with a
 *     zero border of `halo` pixels that producers never write (so 3x3/pad-1 consumers need no
 *     bounds checks).  The caller zeroes a buffer once at allocation.
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: yolo_igemm_desc gained tile_px / split_slabs, yolo_wgrad_desc gained dw_sumsq / slabs / slab_floats, the multi-tensor Adam entries
 *    gained skip_flag.  A binding must refuse a library whose version differs from the header it was
 *    written against (yolo/_hip.py does): the descriptors are passed by pointer and a shorter struct would be read past its end. */
#define YOLO_HIP_ABI_VERSION 2

#define YOLO_E_ARG (-1)         /* bad argument (null pointer, size out of range)              */
#define YOLO_E_UNSUPPORTED (-2) /* shape not supported by the kernels (see each function)      */

typedef void *yolo_stream_t;

int yolo_hip_abi_version(void);
const char *yolo_hip_last_error(void);
/* Diagnostics (libyolo_hip_diag.so, `make diag`, -DIGEMM_STAMPS): the staggered yolo_igemm configurations write eight
 * s_memtime stamps of K iteration `k_iter` per wave of the first 512 workgroups to buf[(workgroup * 8 + wave) * 8 + i]
 * (int64, device memory of 512 * 64 entries; NULL switches it off).  The product library compiles no stamps and returns
 * YOLO_E_UNSUPPORTED for a non-NULL buffer.  tools/stamps_igemm.py reads them. */
int yolo_debug_stamps(void *buf, int k_iter);
/* A non-blocking HIP stream on the current device: of the lowest scheduling priority the device offers (low != 0) or of the
 * default one.  For work off the critical path (weight gradients beside the data-gradient chain, a deferred optimizer pass):
 * the dispatcher then prefers the main chain's workgroups.  The stream lives until the process ends. */
int yolo_stream_create(int low, yolo_stream_t *out);

/* ---------------------------------------------------------------------------------------------
 * Post-processing: S x S x B box decode, pairwise IoU, per-image NMS.  fp64 arithmetic on fp32
 * inputs in the reference's operation order -> bit-exact records, class ids and kept indices.
 * ------------------------------------------------------------------------------------------- */

/* Replaces YOLOInference.parse_predictions (src/yolo/inference.py:170-210) and
 * mAPMetric._parse_predictions (src/yolo/metrics.py:185-218) for a whole batch.
 *   pred    [N][S][S][5B+C] fp32
 *   rec     [N][S*S*B][6] f64 = {class_id, conf*prob, x, y, w, h}; per image the first counts[n]
 *           rows are valid, in (row i, col j, box b) scan order
 *   counts  [N] int32
 * Limits: S*S*B <= 1024, B <= 8. */
int yolo_decode(const float *pred, int N, int S, int B, int C, double conf_thr,
                double *rec, int32_t *counts, yolo_stream_t stream);

/* Replaces mAPMetric._parse_ground_truth (src/yolo/metrics.py:232-256).
 *   rec [N][S*S][5] f64 = {class_id, x, y, w, h}; counts [N]. */
int yolo_decode_gt(const float *tgt, int N, int S, int B, int C,
                   double *rec, int32_t *counts, yolo_stream_t stream);

#define YOLO_NMS_INFERENCE 0 /* inference.py:212-249,298-317: IoU has +1e-6, output in confidence order   */
#define YOLO_NMS_METRICS 1   /* metrics.py:258-341: union==0 -> 0, no eps, output grouped by class         */

/* Replaces YOLOInference.non_max_suppression (src/yolo/inference.py:298-317, variant 0) and
 * mAPMetric._apply_nms (src/yolo/metrics.py:270-296, variant 1), one 4-wave workgroup per image (ballot-built
 * suppression matrix, bit-mask greedy sweep).
 *   rec [N][max_per_img][6] f64 as written by yolo_decode; counts [N]
 *   keep [N][max_per_img] int32: indices into the image's records in the reference's OUTPUT order;
 *   keep_counts [N].
 * A later box survives a kept one iff class differs or IoU < thr (IoU == thr suppresses).
 * Limit: max_per_img <= 1024 (<= 128: suppression matrix by ballots in one pass; above, e.g. S = 14, B = 3 -> 588 boxes:
 * row-by-row sweep with a 1024-bit mask -- same comparisons, same result). */
int yolo_nms(const double *rec, const int32_t *counts, int N, int max_per_img, double thr, int variant,
             int32_t *keep, int32_t *keep_counts, yolo_stream_t stream);

/* Pairwise IoU matrix out[na][nb] between boxes a[na][4], b[nb][4] (x,y,w,h f64); the scalar
 * formulas of YOLOInference.iou (inference.py:229-249, variant 0) and mAPMetric._calculate_iou
 * (metrics.py:313-341, variant 1). */
int yolo_pairwise_iou(const double *a, int na, const double *b, int nb, int variant,
                      double *out, yolo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Loss.  Replaces YOLOLoss.forward + compute_iou (src/yolo/loss.py:87-212) AND its autograd
 * backward in one pass: responsible-box selection by IoU, the five partial sums, dL/dpred
 * including the gradient that flows through the (non-detached) IoU target.
 *   pred,tgt [N][S][S][5B+C] fp32
 *   out      [8] fp32 : {total, coord, conf_obj, conf_noobj, class} each already / N, then
 *            out[5] = error flag (1.0 if some object cell selects a target slot >= B; the
 *            reference raises IndexError there), out[6..7] unused
 *   dpred    [N][S][S][5B+C] fp32 = d total / d pred, or NULL (forward only)
 *   work     [8*N] f64 caller workspace (per-image partial sums; deterministic reduction order)
 * Limits: B <= 8, S*S <= 1024. */
int yolo_loss_fwd_bwd(const float *pred, const float *tgt, int N, int S, int B, int C,
                      float lambda_coord, float lambda_noobj,
                      float *out, float *dpred, double *work, yolo_stream_t stream);

/* YOLOLoss.compute_iou (src/yolo/loss.py:174-212) for n box pairs, fp32. */
int yolo_loss_iou(const float *boxes1, const float *boxes2, long n, float *out, yolo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Network layers.  bf16 storage, fp32 MFMA accumulation.  These replace the aten kernels behind
 * nn.Conv2d + nn.LeakyReLU(0.1), nn.MaxPool2d(2,2), nn.Linear as instantiated by
 * YOLOv1Backbone (src/yolo/models.py:47-84), the FC head (models.py:239-245) and DetectionHead
 * (models.py:313-332).
 * ------------------------------------------------------------------------------------------- */

/* One implicit-GEMM problem:  out[px][co] = epi( sum_{tap,c} in[row(px) + tapoff(tap) + c] * w[co][tap][c] )
 * It covers conv forward, conv data-gradient (weights packed transposed+flipped), and Linear
 * forward / data-gradient (a 1x1 conv on a 1x1 image).  All "elements" are bf16 elements. */
typedef struct yolo_igemm_desc {
    /* output pixel grid */
    int32_t N, Ho, Wo;
    /* input addressing: row(px=(n,oy,ox)) = n*in_img_stride + oy*in_row_stride*sy + ox*in_px_stride*sx ... */
    int64_t in_img_stride;  /* elements between images of the (padded) input                       */
    int32_t in_row_stride;  /* elements between input rows                                         */
    int32_t in_px_stride;   /* elements between input pixels (= channel count of the buffer)       */
    int32_t in_off;         /* element offset of (row 0, px 0, tap 0) inside an image              */
    int32_t stride;         /* conv stride (1 or 2)                                                */
    int32_t KH, KW;         /* taps                                                                */
    int32_t tap_len;        /* contiguous input elements per tap (= Cin; multiple of BK)           */
    int32_t Cout;           /* output channels (rows of w)                                         */
    /* output addressing (padded NHWC): out[(n*out_img_stride) + (oy*out_row_stride) + ox*out_px_stride + out_off + co] */
    int64_t out_img_stride;
    int32_t out_row_stride, out_px_stride, out_off;
    /* epilogue */
    int32_t epilogue;       /* YOLO_EPI_*                                                          */
    float slope;            /* LeakyReLU negative slope                                            */
    int32_t out_fp32;       /* 1: out is fp32, 0: bf16                                             */
    int32_t split_k;        /* >1: K split over blockIdx.y, fp32 atomicAdd into out (out_fp32=1,
                               epilogue NONE; caller zero-fills out and applies bias afterwards)   */
    /* aux addressing for YOLO_EPI_MUL_DLRELU (same form as the output addressing) */
    int64_t aux_img_stride;
    int32_t aux_row_stride, aux_px_stride, aux_off;
    int32_t pool2;          /* 1: fuse MaxPool2d(2,2) into the epilogue (conv -> LeakyReLU -> pool, models.py:49-55):
                               Ho/Wo stay the CONV output size, out_* address the pooled map [Ho/2][Wo/2];
                               2: as 1, and the UN-pooled activation is also written, to `aux` with the aux_* strides
                               (training keeps it for the backward pass);
                               3: as 1, and `aux` receives the ARG-MAX CODES instead: uint16 per (pooled pixel, 8 channels)
                               at index (pooled element address + channel) / 8 -- a buffer of the pooled map's geometry
                               with an eighth of its elements -- 2 bits per channel = window position 2*dy + dx of the
                               first maximum of the activations as stored.  With the pooled map that is all the backward
                               pass needs (yolo_maxpool2_bwd_codes): the un-pooled activation is never written       */
    int32_t w_blocked;      /* 1: w is in the panel layout of yolo_pack_fc_weight_blocked (Linear layers)      */
    int32_t tile_order;     /* 0 = heuristic; 1 = channel tiles fastest; 2 = pixel tiles fastest inside an XCD's range   */
    int32_t tile_hint;      /* 0 = let the library pick the tile configuration; 1: 128x128, 2: 256x128
                               (8 waves, 3 stages), 3: 128x64, 4: 64x128, 5/6: 1/2 on the 16x16x32
                               MFMA shape, 11: 256x128x64 and 12: 256x256x32 / 13: 256x128x32 with the staggered
                               two-phase schedule (8 waves), 14: 256 x 208 x 32 staggered with an uneven 7 / 6 column
                               split between the wave groups (see tile_px), 15: the same tile with the register-pipelined
                               one-barrier loop (even number >= 4 of 32-deep K steps per split; no pool2 / bn_stats /
                               atomics), 7-10: BK = 32 variants of 64x128, 19: streaming 1x1 convolution of the thin-K pointwise
                               layers (1x1, tap_len 64 / 128 / 192 / 256 / 512, Cout % 64 == 0, bf16 out, N*Ho*Wo % 16 == 0: the weight
                               panel sits in LDS, every wave walks 16-pixel groups with the next group's activation fragments and
                               the residual vectors in flight; igemm_stream.hip), 20 / 21: the persistent kernels (igemm_persist.hip: one software
                               pipeline over all tiles of a workgroup, tiles of 256 x 208 / 224 drawn from per-XCD queues, epilogue out of the
                               accumulator registers), 22: 3x3 / stride-1 conv of 64 -> 64 channels on maps of (16k) x (16k) pixels with the weight
                               panel resident in LDS and the input patch of a 16 x 16 tile staged once (conv_c64.hip; epilogue NONE / BIAS /
                               BIAS_LRELU, bf16 out, bn_stats allowed, no pool2 / split_k)  (tuning / tests;
                               a caller that wants the best plan times them per problem, as engine.igemm_call does) */
    int64_t px_begin, px_end; /* compute only output pixels [px_begin, px_end) of the flattened (n, oy, ox) index
                               (0, 0 = all).  Lets a caller run the bulk of a layer with a large tile in whole
                               rounds of the 256 CUs and the remainder with a small tile in one short round,
                               instead of a last round that keeps a few CUs busy for a full large-tile time */
    int32_t skew_phases;    /* > 1 (8-wave configurations, launches of more than 256 workgroups): the first-round
                               workgroups start (slot % skew_phases) * skew_step shader cycles late, so that the CUs
                               do not all reach their output stores at the same moment (0 = off)                  */
    int32_t skew_step;
    void *bn_stats;         /* != NULL (bf16 output, no split_k / pool2 / pixel range): YOLO_BN_ACC_REPLICAS * 2*Cout doubles into
                               which the launch adds, per output channel, the sum and the sum of squares of the values it
                               stored -- the statistics pass of a following yolo_batchnorm_train_fwd (stats_ready = 1) */
    int32_t tile_px;        /* > 0: a workgroup's tile covers only tile_px pixels of the flattened (n, oy, ox) index (at most the
                               configuration's pixel-tile edge; its remaining slots idle).  The layers of this network have
                               N * 49 * 4^k output pixels, which never divide into whole rounds of 256 CUs with 128- or 256-pixel
                               tiles; tile_hint 14 (256 channels x 208 pixel slots, uneven staggered split) with tile_px = 196
                               gives 64 * 4^k tiles per channel tile -- whole rounds at batch 64.  0 = all slots         */
    int32_t split_slabs;    /* split_k > 1 only.  1: split s STORES its partial tile into slab s of `out` = fp32
                               [split_k][N*Ho*Wo][Cout] (no zero fill, no atomics) and yolo_igemm_finish adds the slabs in
                               the fixed order 0, 1, .. -- bit-reproducible for any split count.  0: fp32 atomics into one
                               zero-filled [N*Ho*Wo][Cout] buffer                                                       */
} yolo_igemm_desc;

#define YOLO_EPI_NONE 0        /* out = acc                                                        */
#define YOLO_EPI_BIAS 1        /* out = acc + bias[co]                                             */
#define YOLO_EPI_BIAS_LRELU 2  /* out = lrelu(acc + bias[co])            (Conv2d/Linear + LeakyReLU) */
#define YOLO_EPI_MUL_DLRELU 3  /* out = acc * (aux[px][co] > 0 ? 1 : slope)  (dgrad through the
                                  previous layer's LeakyReLU; aux = that layer's output, bf16,
                                  addressed by the aux_* strides)                                                        */

#define YOLO_EPI_BIAS_ADD_LRELU 4 /* out = lrelu(acc + bias[co] + aux[px][co])  (residual add of a ResNet bottleneck,
                                    slope 0 = ReLU; aux = identity / downsample branch, bf16, aux_* strides)  */

int yolo_igemm(const yolo_igemm_desc *d, const void *in_bf16, const void *w_bf16 /*[Cout][KH*KW*tap_len]*/,
               const float *bias, const void *aux_bf16, void *out, yolo_stream_t stream);
/* Finishing pass of a split-K convolution.  Few-pixel, deep-K layers (the 7x7x1024 ones: 3136 pixels at batch 64,
 * K = 9216) have too few output tiles to fill 256 CUs; the caller runs yolo_igemm with split_k > 1, out_fp32 = 1,
 * YOLO_EPI_NONE into a zero-filled dense fp32 [N*Ho*Wo][Cout] buffer (or, with split_slabs = 1, into split_k slabs of
 * that shape) and then this pass, with the layer's REAL descriptor (epilogue, out_* / aux_* strides; split_k = the number
 * of slabs to add in fixed order when split_slabs = 1, else ignored), applies bias / LeakyReLU / LeakyReLU'(aux) and
 * writes bf16. */
int yolo_igemm_finish(const yolo_igemm_desc *d, const float *acc, const float *bias, const void *aux_bf16,
                      void *out_bf16, yolo_stream_t stream);

/* Weight-gradient of a conv / Linear, "flat" pixel indexing:
 *     dw[co][tap][ci] (+)= sum_{p < P} dy[p*dy_px_stride + co] * x[p*x_px_stride + tapoff(tap) + ci]
 *     tapoff(ky,kx) = (ky - pad)*x_row_stride + (kx - pad)*x_px_stride
 * p runs over EVERY pixel slot of the zero-haloed dy buffer (halo slots hold zeros and add nothing),
 * so dy and x must share one padded geometry (stride-1 convs; a stride-2 conv passes a zero-stuffed
 * dy of the input's geometry).  x needs a guard band of finite values (zeros) of at least
 * |tapoff| elements before and after the buffer.  dw is fp32 in the packed layout
 * [Cout][KH*KW][Cin]; with split > 1 or accumulate != 0 it is accumulated with fp32 atomics
 * (caller zero-fills / owns the previous value).  db[co] += sum_p dy[p][co] if db != NULL
 * (always accumulated: caller zero-fills).  Replaces the weight / bias gradient of aten
 * convolution_backward and addmm backward. */
typedef struct yolo_wgrad_desc {
    int64_t P;                       /* dy pixel slots                                            */
    int32_t dy_px_stride, x_px_stride; /* elements between pixel slots (multiples of 8)           */
    int32_t Cout, Cin;
    int32_t KH, KW, pad;
    int64_t x_row_stride;            /* elements between input rows (multiple of 8)               */
    int32_t split;                   /* pixel-range split (uniform, 2-D grid).  0: the library's two-segment
                                        schedule -- whole rounds of the chip's workgroup slots plus a finer-split
                                        tail round; the caller zero-fills dw (tiles that are split over
                                        several workgroups are accumulated with atomics, the others stored) */
    int32_t accumulate;              /* 1: add into dw even when split == 1                       */
    int32_t variant;                 /* 0: choose; 1: 128x128 tile, 4 waves, 2 stages; 2: 256x128 tile, 8 waves, 3 stages;
                                        3: 2 + staggered two-phase schedule; 4: 128x128 tile, 8 waves; 5: 256x256 tile, wave
                                        tile 128x64, four 32-pixel stages, register-pipelined one-barrier loop
                                        (wgrad_pipe.hip; Cin in {64, 128}: 256 / Cin taps per tile); 6: the tiles and schedule of
                                        5 with FOUR waves of 128x128, one per SIMD, 256 accumulator registers in AGPRs
                                        (wgrad_wide.hip)  (tests / tuning; all agree bit for bit with one pixel range per tile)  */
    /* Pixel geometry (geo_W == 0: "flat" indexing, p IS the slot).  Otherwise the reduction runs over the P = N*geo_H*geo_W
     * pixels p = (n*geo_H + oy)*geo_W + ox only, and pixel p lives in slot
     *     n*geo_img_slots + oy*geo_row_slots + ox*geo_px_slots + geo_slot0
     * of BOTH buffers (dy at slot*dy_px_stride, x at slot*x_px_stride + tap offset): the interior of a zero-haloed buffer
     * (geo_px_slots = 1) skips the halo slots -- 1.04x .. 1.65x fewer K steps -- and geo_px_slots = 2 with
     * geo_row_slots = 2 * row pitch visits only the non-zero slots of a stride-2 conv's zero-stuffed gradient (4x fewer). */
    int32_t geo_W, geo_H, geo_img_slots, geo_row_slots, geo_px_slots, geo_slot0;
    double *dw_sumsq;                /* optional (device double, NULL = off): += sum of squares of the dw this launch stores -- the
                                        global gradient norm of clip_grad_norm_ (trainer.py:79) then needs no pass over the 822 MB
                                        gradient of the Linear behind nn.Flatten.  Only with every tile stored by one workgroup
                                        (split = 1, accumulate = 0, variant 0 / 1, Cin % 4 == 0); otherwise YOLO_E_UNSUPPORTED */
    float *slabs;                    /* optional (variant 5 / 6, accumulate = 0, Cin % 4 == 0): scratch of slab_floats floats.  Every workgroup then
                                        STORES its 256 x 256 partial tile there and a second kernel adds the partials of a tile in pixel-range
                                        order into dw -- no fp32 atomics on dw (1.5 TB/s chip-wide, 60-90 k cycles per workgroup), dw need not
                                        be zero-filled, and dw is bit-reproducible.  yolo_wgrad_slab_floats gives the size a launch needs */
    int64_t slab_floats;
} yolo_wgrad_desc;

/* floats of yolo_wgrad_desc.slabs the launch described by d would use (0: the variant has no slab mode); launches nothing */
int yolo_wgrad_slab_floats(const yolo_wgrad_desc *d, long *floats);
int yolo_wgrad(const yolo_wgrad_desc *d, const void *x_bf16, const void *dy_bf16,
               float *dw_packed, float *db, yolo_stream_t stream);

/* MaxPool2d(2,2) on padded NHWC bf16 (models.py:51,55,65,72) and its backward fused with the
 * backward of the LeakyReLU that precedes the pool:
 *   dz[n][y][x][c] = (y,x is the arg-max of its 2x2 window, first in scan order on ties)
 *                    ? dpool[n][y/2][x/2][c] * (yfull > 0 ? 1 : slope) : 0 */
typedef struct yolo_pool_desc {
    int32_t N, H, W, C;       /* un-pooled logical size                                          */
    int32_t in_halo, out_halo;
} yolo_pool_desc;
/* MaxPool2d(3, stride 2, pad 1) forward (ResNet stem).  The input must be >= 0 (it follows a ReLU), so
 * the zero halo of the buffer is equivalent to the -inf padding of the operator.  H, W = input size. */
int yolo_maxpool3s2_fwd(const yolo_pool_desc *d, const void *x_bf16, void *y_bf16, yolo_stream_t stream);
/* its backward (aten max_pool2d_with_indices_backward): dx[n][h][w][c] = sum of dy over the windows whose arg-max
 * (first maximum in row-major order, recomputed from x) is (h, w).  d->out_halo = halo of dy, dx_halo = halo of dx. */
int yolo_maxpool3s2_bwd(const yolo_pool_desc *d, const void *x_bf16, const void *dy_bf16, void *dx_bf16, int dx_halo,
                        yolo_stream_t stream);
int yolo_maxpool2_fwd(const yolo_pool_desc *d, const void *x_bf16, void *y_bf16, yolo_stream_t stream);
int yolo_maxpool2_bwd_lrelu(const yolo_pool_desc *d, const void *yfull_bf16, const void *dpool_bf16,
                            float slope, void *dz_bf16, yolo_stream_t stream);
/* The same gradient from the POOLED activation (geometry of dpool) and the arg-max codes a fused conv + pool epilogue left
 * (yolo_igemm pool2 = 3, yolo_conv_stem7_fwd codes): codes[(pooled element address) / 8] = uint16, 2 bits per channel.  The
 * un-pooled activation is not read -- training never stores it (models.py:49-55: conv -> LeakyReLU -> MaxPool2d). */
int yolo_maxpool2_bwd_codes(const yolo_pool_desc *d, const void *ypool_bf16, const void *codes_u16, const void *dpool_bf16,
                            float slope, void *dz_bf16, yolo_stream_t stream);

/* ---- layout / precision conversion at the boundary to PyTorch-layout fp32 tensors ------------- */

/* NCHW fp32 -> padded NHWC bf16 with Cpad >= C channels (extra channels zero). */
int yolo_nchw_f32_to_nhwc_bf16(const float *x, int N, int C, int H, int W, void *y, int Cpad,
                               int halo_lo, int halo_hi, yolo_stream_t stream);
/* padded NHWC bf16 -> NCHW fp32 (module boundary, e.g. YOLOv1Backbone.forward's return value). */
int yolo_nhwc_bf16_to_nchw_f32(const void *x, int N, int C, int H, int W, int halo, float *y, yolo_stream_t stream);
/* padded NHWC bf16 -> NCHW bf16: nn.Flatten order (c*H*W + h*W + w) in front of the FC head
 * (src/yolo/models.py:240), so that Linear weights keep their state_dict column order. */
int yolo_nhwc_bf16_to_nchw_bf16(const void *x, int N, int C, int H, int W, int halo, void *y, yolo_stream_t stream);

/* OIHW fp32 -> packed bf16 [Cout][KH][KWp][Cinp] (forward operand) and, if wt != NULL,
 * [Cin][KH][KW][Cout] with taps flipped (data-gradient operand).  KWp/Cinp >= KW/Cin pad with zeros. */
int yolo_pack_conv_weight(const float *w_oihw, int Cout, int Cin, int KH, int KW, int Cinp, int KWp,
                          void *w_fwd_bf16, void *w_dgrad_bf16, yolo_stream_t stream);
/* Linear weight [O][K] fp32 -> bf16 [O][K'] with the K axis permuted from (c, hw) to (hw, c) order
 * (HW = 1: plain cast) and, if wt != NULL, the transposed copy [K'][O] for the data-gradient. */
int yolo_pack_fc_weight(const float *w, int O, int C, int HW, void *w_fwd_bf16, void *w_t_bf16, yolo_stream_t stream);
/* Linear weight [O][K] fp32 (K % 64 == 0) -> bf16 panels [ceil(O/128)][K/64][128][64], rows >= O zero.
 * With yolo_igemm_desc.w_blocked = 1 each LDS stage of the weight stream is one contiguous 16-KB read
 * (a Linear layer at batch 64 is HBM-bound on its 822 MB / 411 MB weight). */
int yolo_pack_fc_weight_blocked(const float *w, int O, long K, void *w_panels_bf16, yolo_stream_t stream);
/* The same panels with the K axis permuted from (c, hw) -- nn.Flatten of an NCHW map (models.py:239) -- to (hw, c): the Linear layer
 * then reads a dense NHWC conv output as it lies in memory and the flatten pass disappears (inference). */
int yolo_pack_fc_weight_blocked_hwc(const float *w, int O, int C, int HW, void *w_panels_bf16, yolo_stream_t stream);
/* packed fp32 gradient [Cout][KH][KWp][Cinp] -> OIHW fp32 (accumulate=0: overwrite, 1: add). */
int yolo_unpack_conv_wgrad(const float *dw_packed, int Cout, int Cin, int KH, int KW, int Cinp, int KWp,
                           float *dw_oihw, int accumulate, yolo_stream_t stream);
/* Forward of the 7x7 / stride-2 / pad-3 stem conv (models.py:49-51): x = NHWC4 bf16 with halo 3, w = the packed
 * [64][7][8][4] bf16 panel of yolo_pack_conv_weight(Cinp = 4, KWp = 8), out = NHWC bf16 with 64 channels
 * (out_off = element offset of output pixel (0,0), pixel stride 64).  y = lrelu(conv + bias, slope); pool2 = 1 also
 * applies MaxPool2d(2,2) and out_* then address the POOLED map.  The raw input patch of an 8 x 16 output tile is staged
 * once in LDS and the 4x overlap of neighbouring pixels' 7 x 8 windows is resolved by the MFMA operand read addresses
 * (the generic yolo_igemm gathers every window from L2: 448 B per output pixel).  Ho % 8 == 0, Wo % 16 == 0.
 * out_full (pool2 = 1 only, may be NULL): the un-pooled activation is written as well -- training needs it for the
 * backward of the pool and of the LeakyReLU -- so that the separate pooling pass disappears there too.
 * pool2 = 3: out_full receives the ARG-MAX CODES instead (uint16 per pooled pixel and 8 channels at (pooled element address) / 8,
 * 2 bits per channel, as yolo_igemm pool2 = 3; full_* unused): with the pooled map that is all the backward pass needs. */
int yolo_conv_stem7_fwd(const void *x_nhwc4_bf16, const void *w_packed_bf16, const float *bias, int N, int Ho, int Wo,
                        long x_img_stride, int x_row_stride, float slope, int pool2, void *out_bf16,
                        long out_img_stride, int out_row_stride, int out_off, void *out_full_bf16,
                        long full_img_stride, int full_row_stride, int full_off, yolo_stream_t stream);
/* The same from the caller's NCHW fp32 image batch [N][3][H][W] (H, W even; Ho = H/2, Wo = W/2): the input patch of a tile is
 * converted to NHWC4 bf16 on its way into LDS (same rounding as yolo_nchw_f32_to_nhwc_bf16), so the separate layout pass and its
 * 103 MB intermediate (batch 64) disappear.  Same results bit for bit. */
int yolo_conv_stem7_fwd_f32(const float *x_nchw_f32, const void *w_packed_bf16, const float *bias, int N, int H, int W, float slope,
                            int pool2, void *out_bf16, long out_img_stride, int out_row_stride, int out_off, void *out_full_bf16,
                            long full_img_stride, int full_row_stride, int full_off, yolo_stream_t stream);
/* Weight + bias gradient of the 7x7 / stride-2 / pad-3 stem conv (models.py:49; 3 input channels stored
 * NHWC4 with halo 3, Cout = 64) WITHOUT the unfolded copy: the unfolding happens in the LDS read addresses of
 * the MFMA operands.  dy: NHWC bf16 with 64 channels, dy_off = element offset of output pixel (0,0).
 * Ho % 8 == 0 and Wo % 16 == 0 required (otherwise: yolo_im2col_rows + yolo_wgrad).  Writes (does not
 * accumulate) dw in OIHW fp32 [64][3][7][7] and db[64] (NULL: skipped); deterministic: per-workgroup
 * partials in `scratch` (>= 14400 floats; 768 * 14400 for full speed) are summed in a fixed order. */
int yolo_wgrad_stem7(const void *x_nhwc4_bf16, const void *dy_bf16, int N, int Ho, int Wo, long x_img_stride,
                     int x_row_stride, long dy_img_stride, int dy_row_stride, int dy_off, float *dw_oihw,
                     float *db, float *scratch, long scratch_elems, yolo_stream_t stream);
/* The same with the backward of the MaxPool2d(2,2) + LeakyReLU that follow the stem fused in: y_full = the stem's
 * un-pooled activation (64 channels), dpool = gradient of the pooled map [Ho/2][Wo/2]; the gradient tile
 * (arg-max of every window gets dpool * LeakyReLU', the rest 0 -- what yolo_maxpool2_bwd_lrelu writes) is rebuilt in
 * registers per tile, so that 411 MB gradient buffer (batch 64) is neither written nor read.  Same results bit for bit. */
int yolo_wgrad_stem7_pooled(const void *x_nhwc4_bf16, const void *y_full_bf16, int N, int Ho, int Wo, long x_img_stride,
                            int x_row_stride, long y_img_stride, int y_row_stride, int y_off, const void *dpool_bf16,
                            long dp_img_stride, int dp_row_stride, int dp_off, float slope, float *dw_oihw, float *db,
                            float *scratch, long scratch_elems, yolo_stream_t stream);
/* ... and from what a stem forward with pool2 = 3 left instead of the un-pooled activation: y_pooled (geometry of dpool: the
 * dp_* strides address both) and the arg-max codes (uint16 at (pooled element address) / 8).  Reads 154 MB instead of 514 MB at
 * batch 64; same results bit for bit. */
int yolo_wgrad_stem7_codes(const void *x_nhwc4_bf16, const void *y_pooled_bf16, const void *codes_u16, int N, int Ho, int Wo,
                           long x_img_stride, int x_row_stride, const void *dpool_bf16, long dp_img_stride, int dp_row_stride,
                           int dp_off, float slope, float *dw_oihw, float *db, float *scratch, long scratch_elems,
                           yolo_stream_t stream);
/* Whole-model forms of the two calls above: every conv layer of the model in ONE launch (LDS-tiled,
 * all HBM accesses in runs of >= 128 B).  Layers need Cout % 64 == 0 (unpack: % 4), Cin % 64 == 0,
 * KH*KW <= 9 and no padding (Cinp = Cin, KWp = KW); either output of a pack item may be NULL. */
#define YOLO_PACK_MAX 32
typedef struct yolo_conv_pack_item {
    const float *w;            /* OIHW fp32 */
    void *w_fwd_bf16;          /* [Cout][KH][KW][Cin] */
    void *w_dgrad_bf16;        /* [Cin][KH][KW][Cout], taps flipped */
    int Cout, Cin, KH, KW;
} yolo_conv_pack_item;
typedef struct yolo_conv_unpack_item {
    const float *dw_packed;    /* [Cout][KH][KW][Cin] fp32 */
    float *dw_oihw;
    int Cout, Cin, KH, KW;
} yolo_conv_unpack_item;
int yolo_pack_conv_weights_multi(const yolo_conv_pack_item *items, int count, yolo_stream_t stream);
int yolo_unpack_conv_wgrads_multi(const yolo_conv_unpack_item *items, int count, yolo_stream_t stream);
/* Row-segment unfold for the 7x7/stride-2 first layer's WEIGHT GRADIENT only (Cin=3 gives no
 * channel-contiguous K axis): xcol[n][oy+h][ox+h][ky*seg + j] = x[n][oy*stride+ky][(ox*stride)*px + j],
 * written in the zero-haloed geometry of the layer's output so that yolo_wgrad's flat indexing
 * applies with one tap.  (models.py:49; the forward conv of that layer is im2col-free.) */
int yolo_im2col_rows(const void *x_bf16, long x_img_stride, int x_row_stride, int x_px_stride, int stride,
                     int KH, int seg, int N, int Ho, int Wo, int out_halo, void *xcol_bf16, yolo_stream_t stream);
/* fp32 [R][Ccols] -> bf16 transposed y[c*ld + r] (ld >= R; columns R..ld-1 are not written: the
 * caller zero-fills once).  Data-gradient operand of a Linear layer ([K][O] from [O][K]). */
int yolo_transpose_f32_to_bf16(const float *x, int R, int Ccols, void *y_bf16, int ld, yolo_stream_t stream);
/* bf16 [R][ldx] -> bf16 transposed y[c*ldy + r] (columns R..ldy-1 of y are not written). */
int yolo_transpose_bf16(const void *x_bf16, int R, int Ccols, int ldx, void *y_bf16, int ldy, yolo_stream_t stream);
/* Data-gradient of the Linear layer behind nn.Flatten (models.py:240-241), computed TRANSPOSED by
 * yolo_wgrad -- dxT[k][n] = sum_o W[o][k] * g[n][o] with the forward bf16 copy of W as the "dy" operand, so
 * that no [K][O] transposed copy of the 205 M-element weight is ever written -- and finished here:
 * k = c*H*W + h*W + w  ->  zero-haloed NHWC bf16 gradient, times LeakyReLU'(y_act) (y_act NULL: none). */
int yolo_fc_dgrad_to_nhwc(const float *dxT, int N, int C, int H, int W, int halo, const void *y_act_bf16,
                          float slope, void *g_bf16, yolo_stream_t stream);
int yolo_cast_f32_to_bf16(const float *x, long n, void *y_bf16, yolo_stream_t stream);
int yolo_cast_bf16_to_f32(const void *x_bf16, long n, float *y, yolo_stream_t stream);
/* y = lrelu(x + bias[col]) over [R][Ccols] fp32, bf16 and/or fp32 outputs (finishes a split-K Linear). */
int yolo_bias_lrelu_rows(const float *x, const float *bias, int R, int Ccols, float slope,
                         void *y_bf16, float *y_f32, yolo_stream_t stream);
/* the same finishing pass for a Linear layer whose split_k workgroups STORED their partial results as slabs
 * (yolo_igemm_desc.split_slabs): x = fp32 [slabs][R][Ccols], added in the fixed order 0, 1, .. -> bit-reproducible forward. */
int yolo_bias_lrelu_rows_slabs(const float *x, int slabs, const float *bias, int R, int Ccols, float slope,
                               void *y_bf16, float *y_f32, yolo_stream_t stream);

/* y[r][0..ld) = bf16( x[r][c] * (mask ? (mask[r][c] ? scale : 0) : 1) * (act ? (act[r][c] > 0 ? 1 : slope) : 1) ),
 * columns Cc..ld-1 zero.  x fp32 [R][Cc], mask u8 [R][Cc], act bf16 [R][Cc].  Backward of
 * Dropout(0.5) + LeakyReLU(0.1) of the FC head (models.py:242-243) and the fp32->bf16 hand-over of
 * dL/dpred (row padding to a multiple of 8 columns). */
int yolo_scale_rows_to_bf16(const float *x, const unsigned char *mask, float scale, const void *act_bf16, float slope,
                            int R, int Cc, int ld, void *y_bf16, yolo_stream_t stream);
/* nn.Dropout forward on bf16: y = mask ? x * scale : 0 (mask u8 drawn by the caller's RNG). */
int yolo_dropout_bf16(const void *x_bf16, const unsigned char *mask, float scale, long n, void *y_bf16, yolo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step (HBM-bound).  Replaces clip_grad_norm_(max_norm=10) + optim.Adam(lr, weight_decay)
 * of the reference's train step (src/yolo/training/trainer.py:79-95, src/train.py:177-179).
 * ------------------------------------------------------------------------------------------- */
/* *acc += sum(g[i]^2)  (device double, caller zero-fills once per step; one call per tensor). */
int yolo_sumsq_f32(const float *g, long n, double *acc, yolo_stream_t stream);
/* One tensor of torch.optim.Adam (amsgrad=False, L2 weight decay) in a single pass; `step` is the
 * 1-based step count.  If norm_sq != NULL the gradient is first scaled by
 * min(1, max_norm / (sqrt(*norm_sq) + 1e-6)) exactly as clip_grad_norm_ would have.  If p_bf16 != NULL
 * the updated parameter is also written as bf16 (same layout). */
int yolo_adam_step(float *p, const float *g, float *exp_avg, float *exp_avg_sq, long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, long step, const double *norm_sq,
                   float max_norm, void *p_bf16, yolo_stream_t stream);
/* Multi-tensor forms: ONE launch for a whole parameter list (the reference's optimizer is torch's
 * multi-tensor `_foreach` Adam; 48 of this model's 52 tensors are too small to fill the chip alone).
 * Tables are host arrays; they are copied into the kernel arguments, YOLO_MT_MAX tensors per launch.
 * All tensors of one yolo_adam_step_multi call share `step` and the hyper-parameters.
 * skip_flag (device float, NULL: none): when *skip_flag != 0 the launch updates NOTHING -- the reference raises IndexError inside
 * YOLOLoss.forward for a target that selects a box slot >= B (src/yolo/loss.py:112-118), i.e. before any parameter update; here the
 * loss kernel's error word (out[5] of yolo_loss_fwd_bwd) is read by the host only after the step was enqueued, so the update
 * kernels test it on the device. */
#define YOLO_MT_MAX 48
typedef struct yolo_adam_tensor {
    float *p;            /* parameter, updated in place */
    const float *g;      /* gradient */
    float *m, *v;        /* exp_avg, exp_avg_sq */
    void *p_bf16;        /* optional bf16 shadow of p in the same layout (NULL: none) */
    long n;              /* elements */
} yolo_adam_tensor;
int yolo_sumsq_f32_multi(const float *const *g, const long *n, int count, double *acc, yolo_stream_t stream);
int yolo_adam_step_multi(const yolo_adam_tensor *t, int count, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long step, const double *norm_sq, float max_norm, const float *skip_flag,
                         yolo_stream_t stream);
/* The same update as a BACKGROUND pass: `workgroups` (1 .. 256) persistent workgroups of 1024 threads, each holding one CU to itself
 * (they reserve LDS they do not use), walk the elements.  An HBM-bound pass that occupies exactly that many CUs: launched on a
 * second stream it runs beside MFMA-bound kernels on the other CUs -- e.g. the update of the Linear layers (76 % of this model's
 * optimizer bytes, first used at the END of the next forward) beside the next forward's conv stack.  At most YOLO_MT_MAX tensors. */
int yolo_adam_step_multi_bg(const yolo_adam_tensor *t, int count, float lr, float beta1, float beta2, float eps,
                            float weight_decay, long step, const double *norm_sq, float max_norm, const float *skip_flag,
                            int workgroups, yolo_stream_t stream);
/* g *= min(1, max_norm / (sqrt(*norm_sq) + 1e-6))  (stand-alone clip_grad_norm_ for other optimizers). */
int yolo_clip_scale_f32(float *g, long n, const double *norm_sq, float max_norm, yolo_stream_t stream);

/* BatchNorm2d forward in TRAINING mode (batch statistics) + optional residual add + optional ReLU, in place on a
 * zero-haloed NHWC bf16 conv output z [N][H+2h][W+2h][C] (C % 8 == 0).  The reference's default model trains with a
 * frozen ResNet-50 backbone whose BatchNorm layers still run in training mode (trainer.py:49): they normalise with
 * the batch mean / biased variance and update running_mean / running_var (unbiased) with `momentum`, as aten
 * batch_norm(training=True) does.  acc2c: YOLO_BN_ACC_REPLICAS * 2*C doubles, zero on entry and on return (scratch shared
 * by all layers; the workgroups' partial sums are spread over the replicas because same-address fp64 atomics serialise);
 * scale_shift: 2*C floats of scratch; residual: NHWC bf16 of the same H, W, C (NULL: none).  Forward only. */
#define YOLO_BN_ACC_REPLICAS 16
int yolo_batchnorm_train_fwd(void *z_bf16, int N, int H, int W, int C, int halo, const float *gamma, const float *beta,
                             double eps, double momentum, float *running_mean, float *running_var,
                             const void *residual_bf16, int residual_halo, int relu, double *acc2c,
                             float *scale_shift, void *out_bf16, int out_halo, float *save_mean_invstd,
                             int stats_ready, yolo_stream_t stream);
/* (stats_ready = 1: acc2c already holds the sums -- the producing yolo_igemm accumulated them (yolo_igemm_desc.bn_stats) --
 *  and the statistics pass over z is skipped; stats_ready = 2: z is normalised with running_mean / running_var as they are
 *  (aten batch_norm(training=False): eval() mode of the reference's trunk, src/yolo/models.py:131-176, when gradients flow
 *  through it); nothing is accumulated or updated, save_mean_invstd receives the running mean and 1/sqrt(running_var + eps);
 *  out_bf16 != NULL: the result goes to that buffer [N][H+2*out_halo][W+2*out_halo][C] and z is kept -- a trainable
 *  trunk needs z for the backward pass; save_mean_invstd != NULL: 4*C floats -- batch mean, 1/sqrt(var + eps), and the
 *  scale / shift the forward applied, y = fma(z, scale, shift).)
 *
 * BatchNorm2d backward (training mode) for a conv -> BN [-> + residual] [-> ReLU] unit of a TRAINABLE ResNet trunk
 * (the reference's default run: ResNetBackbone(pretrained=True, freeze=False), src/train.py:144; aten
 * native_batch_norm_backward + threshold_backward).  dy: gradient wrt the unit's output; y: that output (ReLU mask,
 * NULL when the unit has no ReLU, or with relu_from_z = 1: the unit is conv -> BN -> ReLU without a residual and the mask is
 * recomputed as fma(z, scale, shift) > 0, the forward's own expression, which saves reading y twice);
 * z: the conv output the forward normalised; mean_invstd: the 4*C floats saved by the forward.
 *   dy' = dy * [y > 0];  dbeta = sum dy';  dgamma = sum dy' * xhat;  dz = gamma * invstd * (dy' - dbeta/M - xhat * dgamma/M)
 * dz is written at dz[n*dz_img_stride + y*dz_row_stride + x*dz_px_stride + dz_off + c] (doubled strides put it
 * zero-stuffed on the input grid of a stride-2 conv, the form yolo_wgrad / the data gradient read); store_masked_dy: dy'
 * replaces dy in place (the identity branch of a bottleneck receives it).  relu_from_z bit 1 (value 2 or 3): the forward ran
 * with stats_ready = 2 (running statistics): dgamma / dbeta as above with the saved mean / invstd, dz = gamma * invstd * dy'.  acc2c: as above (YOLO_BN_ACC_REPLICAS * 2*C
 * doubles, zero on entry and on return); coef3c: 3*C floats of scratch. */
int yolo_batchnorm_bwd(void *dy_bf16, int dy_halo, const void *y_bf16, int y_halo, const void *z_bf16, int z_halo,
                       int N, int H, int W, int C, const float *gamma, const float *mean_invstd, void *dz_bf16,
                       long dz_img_stride, long dz_row_stride, long dz_px_stride, long dz_off, int store_masked_dy,
                       int relu_from_z, float *dgamma, float *dbeta, double *acc2c, float *coef3c, yolo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * TP / FP matching of mAPMetric on the device.  Replaces the per-class greedy matching loops of
 * src/yolo/metrics.py:343-442 (AP per class and threshold), :444-491 (overall precision / recall) and :568-651 (size
 * buckets), which walk every prediction in Python.  Inputs are the device outputs of yolo_decode + yolo_nms (metrics
 * variant) and yolo_decode_gt of the same batch.  Per kept prediction k of image n (NMS output order):
 *   tp_bits[n*max_per_img + k], bit v*(T+1) + t = 1 iff the prediction is a true positive at thresholds[t] against
 *   ground-truth set v (0 all, 1 small, 2 medium, 3 large: area < small_area / < medium_area / else); t == T is
 *   `extra_threshold` (the reference's overall precision / recall use 0.5 whatever the threshold list is).
 * gt_bucket[n*max_gt + g] = 1 / 2 / 3.  thresholds is a HOST array.  T <= 15, max_per_img <= 128, max_gt <= 64. */
int yolo_map_match(const double *rec, const int32_t *keep, const int32_t *keep_counts, int N, int max_per_img,
                   const double *gt_rec, const int32_t *gt_counts, int max_gt, const double *thresholds, int T,
                   double extra_threshold, double small_area, double medium_area, unsigned long long *tp_bits,
                   int32_t *gt_bucket, yolo_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Image preprocessing.  Replaces YOLOInference.transform / the eval transform of the reference
 * (src/yolo/inference.py:58-66, src/yolo/dataset.py:224-233): Resize((Ho,Wo)) = PIL.Image.resize(BILINEAR) ->
 * ToTensor (/255) -> Normalize(mean, std), from decoded uint8 RGB [N][Hs][Ws][3] in device memory.
 * Resize is Pillow's two-pass 8-bit resampling, bit-exact: hbounds/vbounds = int32 [out][2] (first input index,
 * count), hcoef/vcoef = int32 [out][k] 22-bit fixed-point weights, computed by the host exactly as Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc do (yolo/preprocess.py); tables may be NULL for an axis whose size does
 * not change; tmp = uint8 [N][Hs][Wo][3] scratch for the horizontal pass.  mean3 / std3 are HOST pointers.
 * Outputs (either may be NULL): out_nhwc4 = zero-haloed NHWC4 bf16 [N][Ho+2h][Wo+2h][4] (interior written, channel 3 = 0:
 * the stem's input, = yolo_nchw_f32_to_nhwc_bf16 of the fp32 result), out_nchw = fp32 [N][3][Ho][Wo] (what the
 * reference's transform returns). */
int yolo_preprocess_u8(const unsigned char *src_u8, int N, int Hs, int Ws, int Ho, int Wo, const int32_t *hbounds,
                       const int32_t *hcoef, int hk, const int32_t *vbounds, const int32_t *vcoef, int vk,
                       unsigned char *tmp_u8, const float *mean3, const float *std3, void *out_nhwc4_bf16, int halo,
                       float *out_nchw, yolo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */
