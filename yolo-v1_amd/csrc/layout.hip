// Layout / precision conversion kernels at the boundary between PyTorch-layout fp32 tensors
// (NCHW activations, OIHW / [O][K] weights: the state_dict contract of SURVEY.md 8b) and the
// network's internal format (zero-haloed NHWC bf16 activations, K-contiguous bf16 weight panels).
// All HBM-bound; every kernel moves 8-16 B per lane on its contiguous side.
#include <algorithm>

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

// the 3-channel image with W % 4 == 0: four pixels per thread -- three 16-B loads (one per plane), two 16-B stores
__global__ void __launch_bounds__(256) nchw3_to_nhwc4_x4_kernel(const float *__restrict__ x, int N, int H, int W, bf16_t *__restrict__ y, int lo, int hi)
{
    const int W4 = W >> 2;
    const long total = (long)N * H * W4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int w = (int)(idx % W4) * 4;
    const int h = (int)((idx / W4) % H);
    const int n = (int)(idx / ((long)W4 * H));
    const int Wp = W + lo + hi, Hp = H + lo + hi;
    const long plane = (long)H * W;
    const float *src = x + (long)n * 3 * plane + (long)h * W + w;
    const float4 r = *reinterpret_cast<const float4 *>(src), g = *reinterpret_cast<const float4 *>(src + plane), b = *reinterpret_cast<const float4 *>(src + 2 * plane);
    uint4 o0, o1;
    o0.x = (unsigned)f32_to_bf16(r.x) | ((unsigned)f32_to_bf16(g.x) << 16); o0.y = (unsigned)f32_to_bf16(b.x);
    o0.z = (unsigned)f32_to_bf16(r.y) | ((unsigned)f32_to_bf16(g.y) << 16); o0.w = (unsigned)f32_to_bf16(b.y);
    o1.x = (unsigned)f32_to_bf16(r.z) | ((unsigned)f32_to_bf16(g.z) << 16); o1.y = (unsigned)f32_to_bf16(b.z);
    o1.z = (unsigned)f32_to_bf16(r.w) | ((unsigned)f32_to_bf16(g.w) << 16); o1.w = (unsigned)f32_to_bf16(b.w);
    bf16_t *dst = y + (((long)n * Hp + h + lo) * Wp + w + lo) * 4;       // 8-B aligned (pixel = 8 B); 16-B only if (w + lo) is even
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<uint4 *>(dst) = o0;
        *reinterpret_cast<uint4 *>(dst + 8) = o1;
    } else {
        *reinterpret_cast<uint2 *>(dst) = uint2{o0.x, o0.y};
        *reinterpret_cast<uint4 *>(dst + 4) = uint4{o0.z, o0.w, o1.x, o1.y};
        *reinterpret_cast<uint2 *>(dst + 12) = uint2{o1.z, o1.w};
    }
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

// ---- all conv layers of a model in ONE launch ---------------------------------------------------
// The per-layer kernels above index element by element (stride-KH*KW gathers) and a 24-layer model
// needs 48 + 24 launches of them per training step, most too small to fill the chip.  These go
// through LDS tiles instead -- every HBM access is a run of >= 128 B -- and take the whole layer list
// in the kernel arguments.
constexpr int PK_T = 64;          // co x ci tile edge
struct PackItem {
    const float *w;
    bf16_t *wf, *wd;
    int Cout, Cin, KK;            // KK = KH*KW
    int n_ci_tiles;
    int first;                    // first workgroup of this layer
};
struct PackTable {
    PackItem it[YOLO_PACK_MAX];
    int count, total;
};

__global__ void __launch_bounds__(256) pack_conv_multi_kernel(const PackTable tab)
{
    extern __shared__ __attribute__((aligned(16))) bf16_t tile[];  // [64 co][64*KK + 4]
    int li = 0;
    while (li + 1 < tab.count && tab.it[li + 1].first <= (int)blockIdx.x) ++li;
    const PackItem &L = tab.it[li];
    const int b = blockIdx.x - L.first;
    const int ci0 = (b % L.n_ci_tiles) * PK_T, co0 = (b / L.n_ci_tiles) * PK_T;
    const int KK = L.KK, row = PK_T * KK, pitch = row + 4;
    // load: 64 rows of 64*KK contiguous floats
    const int q_per_row = row / 4;
    for (int idx = threadIdx.x; idx < PK_T * q_per_row; idx += 256) {
        const int r = idx / q_per_row, j = idx - r * q_per_row;
        const float4 v = *reinterpret_cast<const float4 *>(L.w + ((long)(co0 + r) * L.Cin + ci0) * KK + 4 * j);
        uint2 o;
        o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        *reinterpret_cast<uint2 *>(tile + r * pitch + 4 * j) = o;
    }
    __syncthreads();
    // forward operand wf[co][tap][ci]: 8 ci (16 B) per lane, 128-B runs
    if (L.wf) {
        for (int idx = threadIdx.x; idx < PK_T * KK * 8; idx += 256) {
            const int c8 = idx & 7, t = (idx >> 3) % KK, r = (idx >> 3) / KK;
            const bf16_t *src = tile + r * pitch + (c8 * 8) * KK + t;
            unsigned short e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = src[k * KK];
            uint4 o = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                       (unsigned)e[6] | ((unsigned)e[7] << 16)};
            *reinterpret_cast<uint4 *>(L.wf + ((long)(co0 + r) * KK + t) * L.Cin + ci0 + c8 * 8) = o;
        }
    }
    // data-gradient operand wd[ci][KK-1-tap][co]: 8 co per lane
    if (L.wd) {
        for (int idx = threadIdx.x; idx < PK_T * KK * 8; idx += 256) {
            const int c8 = idx & 7, t = (idx >> 3) % KK, ci = (idx >> 3) / KK;
            const bf16_t *src = tile + (c8 * 8) * pitch + ci * KK + t;
            unsigned short e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = src[k * pitch];
            uint4 o = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                       (unsigned)e[6] | ((unsigned)e[7] << 16)};
            *reinterpret_cast<uint4 *>(L.wd + ((long)(ci0 + ci) * KK + (KK - 1 - t)) * L.Cout + co0 + c8 * 8) = o;
        }
    }
}

struct UnpackItem {
    const float *dwp;
    float *dw;
    int Cout, Cin, KK;
    int n_ci_tiles;
    int first;
};
struct UnpackTable {
    UnpackItem it[YOLO_PACK_MAX];
    int count, total;
};
constexpr int UP_CO = 4;  // output channels per workgroup

// packed [co][tap][ci] fp32 -> OIHW [co][ci][tap]: workgroup = 4 co x 64 ci x all taps
__global__ void __launch_bounds__(256) unpack_conv_multi_kernel(const UnpackTable tab)
{
    extern __shared__ __attribute__((aligned(16))) float ftile[];  // [4][KK][65]
    int li = 0;
    while (li + 1 < tab.count && tab.it[li + 1].first <= (int)blockIdx.x) ++li;
    const UnpackItem &L = tab.it[li];
    const int b = blockIdx.x - L.first;
    const int ci0 = (b % L.n_ci_tiles) * PK_T, co0 = (b / L.n_ci_tiles) * UP_CO;
    const int KK = L.KK;
    for (int idx = threadIdx.x; idx < UP_CO * KK * PK_T; idx += 256) {
        const int ci = idx & 63, t = (idx >> 6) % KK, c = (idx >> 6) / KK;
        ftile[(c * KK + t) * 65 + ci] = L.dwp[((long)(co0 + c) * KK + t) * L.Cin + ci0 + ci];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < UP_CO * KK * PK_T; idx += 256) {
        const int j = idx % (KK * PK_T), c = idx / (KK * PK_T);
        const int ci = j / KK, t = j - ci * KK;
        L.dw[((long)(co0 + c) * L.Cin + ci0) * KK + j] = ftile[(c * KK + t) * 65 + ci];
    }
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

// the same panels with the K axis permuted from (c, hw) -- nn.Flatten of an NCHW map -- to (hw, c), the order in which a dense NHWC
// map lies in memory: the Linear layer then reads the conv output directly and the flatten pass disappears (inference)
__global__ void pack_fc_blocked_hwc_kernel(const float *__restrict__ w, int O, int C, int HW, bf16_t *__restrict__ wb)
{
    const long K = (long)C * HW, nk = K / 64;
    const long total8 = (long)((O + 127) / 128) * nk * 128 * 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total8) return;
    const int c8 = (int)(idx & 7);
    const int r = (int)((idx >> 3) & 127);
    const long kt = (idx >> 10) % nk;
    const long ct = (idx >> 10) / nk;
    const long o = ct * 128 + r;
    uint4 out = {0u, 0u, 0u, 0u};
    if (o < O) {
        const long kp = kt * 64 + c8 * 8;                 // (hw, c) position of the group's first element; C % 8 == 0: one hw
        const long hw = kp / C, c = kp - hw * C;
        const float *src = w + o * K + c * HW + hw;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long)e * HW];
        out.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        out.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        out.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
        out.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
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

// bf16 [R][ldx] -> bf16 [Cc][ldy] (small matrices: the batch-side operand of a Linear data-gradient)
__global__ void __launch_bounds__(256) transpose_bf16_ld_kernel(const bf16_t *__restrict__ x, int R, int Cc, int ldx, bf16_t *__restrict__ y, int ldy)
{
    __shared__ bf16_t t[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4)
        if (r0 + k < R && c0 + tx < Cc) t[k][tx] = x[(long)(r0 + k) * ldx + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
        if (c0 + k < Cc && r0 + tx < R) y[(long)(c0 + k) * ldy + r0 + tx] = t[tx][k];
}

// Linear data-gradient, produced transposed by yolo_wgrad as dxT[k][n] (k = c*HW + hw, nn.Flatten order),
// -> zero-haloed NHWC bf16 gradient g[n][h][w][c], times LeakyReLU'(y) of the conv in front of nn.Flatten
__global__ void __launch_bounds__(256) fc_dgrad_to_nhwc_kernel(const float *__restrict__ dxT, int N, int C, int H, int W, int halo,
                                                               const bf16_t *__restrict__ yact, float slope, bf16_t *__restrict__ g)
{
    __shared__ float t[64][65];
    const int HW = H * W;
    const int c0 = blockIdx.x * 64, hw = blockIdx.y, n0 = blockIdx.z * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4)
        if (c0 + k < C && n0 + tx < N) t[k][tx] = dxT[((long)(c0 + k) * HW + hw) * N + n0 + tx];
    __syncthreads();
    const int h = hw / W, w = hw - h * W;
    const int Hp = H + 2 * halo, Wp = W + 2 * halo;
    for (int k = ty; k < 64; k += 4) {
        const int n = n0 + k, c = c0 + tx;
        if (n < N && c < C) {
            const long o = (((long)n * Hp + h + halo) * Wp + w + halo) * C + c;
            float v = t[tx][k];
            if (yact && !(bf16_to_f32(yact[o]) > 0.0f)) v *= slope;
            g[o] = f32_to_bf16(v);
        }
    }
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
__global__ void bias_lrelu_rows_kernel(const float *__restrict__ x, int slabs, const float *__restrict__ bias, int R, int Cc, float slope,
                                       bf16_t *__restrict__ yb, float *__restrict__ yf)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)R * Cc;
    if (idx >= total) return;
    const int c = (int)(idx % Cc);
    float v = x[idx];
    for (int s = 1; s < slabs; ++s) v += x[(long)s * total + idx];      // fixed order: independent of which split finished first
    v += bias ? bias[c] : 0.0f;
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
        if (C == 3 && (W & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0)
            hipLaunchKernelGGL(nchw3_to_nhwc4_x4_kernel, dim3(nblk((long)N * H * (W / 4), 256)), dim3(256), 0, STRM(stream), x, N, H, W, (bf16_t *)y, halo_lo, halo_hi);
        else
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

YOLO_API int yolo_pack_conv_weights_multi(const yolo_conv_pack_item *items, int count, yolo_stream_t stream)
{
    if (!items || count < 0) return fail(YOLO_E_ARG, "yolo_pack_conv_weights_multi: bad argument");
    for (int base = 0; base < count; base += YOLO_PACK_MAX) {
        PackTable tab{};
        int kmax = 1, blocks = 0;
        const int n = std::min(YOLO_PACK_MAX, count - base);
        for (int k = 0; k < n; ++k) {
            const yolo_conv_pack_item &e = items[base + k];
            if (!e.w || (!e.w_fwd_bf16 && !e.w_dgrad_bf16) || e.Cout <= 0 || e.Cin <= 0 || e.KH <= 0 || e.KW <= 0)
                return fail(YOLO_E_ARG, "yolo_pack_conv_weights_multi: layer %d: bad descriptor", base + k);
            if ((e.Cout % PK_T) || (e.Cin % PK_T) || e.KH * e.KW > 9)
                return fail(YOLO_E_UNSUPPORTED, "yolo_pack_conv_weights_multi: layer %d: Cout, Cin must be multiples of 64 and KH*KW <= 9 (use yolo_pack_conv_weight)", base + k);
            if (((uintptr_t)e.w | (uintptr_t)e.w_fwd_bf16 | (uintptr_t)e.w_dgrad_bf16) & 15)
                return fail(YOLO_E_UNSUPPORTED, "yolo_pack_conv_weights_multi: layer %d: pointers must be 16-B aligned", base + k);
            PackItem &t = tab.it[k];
            t.w = e.w; t.wf = (bf16_t *)e.w_fwd_bf16; t.wd = (bf16_t *)e.w_dgrad_bf16;
            t.Cout = e.Cout; t.Cin = e.Cin; t.KK = e.KH * e.KW;
            t.n_ci_tiles = e.Cin / PK_T;
            t.first = blocks;
            blocks += t.n_ci_tiles * (e.Cout / PK_T);
            kmax = std::max(kmax, t.KK);
        }
        tab.count = n; tab.total = blocks;
        const size_t lds = (size_t)PK_T * (PK_T * kmax + 4) * sizeof(bf16_t);  // 74 KB for 3x3
        static bool big_lds = false;
        if (!big_lds) {
            if (hipFuncSetAttribute((const void *)pack_conv_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                return fail(YOLO_E_UNSUPPORTED, "yolo_pack_conv_weights_multi: cannot reserve %zu B of LDS", lds);
            big_lds = true;
        }
        hipLaunchKernelGGL(pack_conv_multi_kernel, dim3(blocks), dim3(256), lds, STRM(stream), tab);
        if (int rc = check_launch("yolo_pack_conv_weights_multi")) return rc;
    }
    return 0;
}

YOLO_API int yolo_unpack_conv_wgrads_multi(const yolo_conv_unpack_item *items, int count, yolo_stream_t stream)
{
    if (!items || count < 0) return fail(YOLO_E_ARG, "yolo_unpack_conv_wgrads_multi: bad argument");
    for (int base = 0; base < count; base += YOLO_PACK_MAX) {
        UnpackTable tab{};
        int kmax = 1, blocks = 0;
        const int n = std::min(YOLO_PACK_MAX, count - base);
        for (int k = 0; k < n; ++k) {
            const yolo_conv_unpack_item &e = items[base + k];
            if (!e.dw_packed || !e.dw_oihw || e.Cout <= 0 || e.Cin <= 0 || e.KH <= 0 || e.KW <= 0)
                return fail(YOLO_E_ARG, "yolo_unpack_conv_wgrads_multi: layer %d: bad descriptor", base + k);
            if ((e.Cout % UP_CO) || (e.Cin % PK_T) || e.KH * e.KW > 9)
                return fail(YOLO_E_UNSUPPORTED, "yolo_unpack_conv_wgrads_multi: layer %d: Cout %% 4, Cin %% 64, KH*KW <= 9 required (use yolo_unpack_conv_wgrad)", base + k);
            UnpackItem &t = tab.it[k];
            t.dwp = e.dw_packed; t.dw = e.dw_oihw;
            t.Cout = e.Cout; t.Cin = e.Cin; t.KK = e.KH * e.KW;
            t.n_ci_tiles = e.Cin / PK_T;
            t.first = blocks;
            blocks += t.n_ci_tiles * (e.Cout / UP_CO);
            kmax = std::max(kmax, t.KK);
        }
        tab.count = n; tab.total = blocks;
        const size_t lds = (size_t)UP_CO * kmax * 65 * sizeof(float);
        hipLaunchKernelGGL(unpack_conv_multi_kernel, dim3(blocks), dim3(256), lds, STRM(stream), tab);
        if (int rc = check_launch("yolo_unpack_conv_wgrads_multi")) return rc;
    }
    return 0;
}

YOLO_API int yolo_transpose_f32_to_bf16(const float *x, int R, int Cc, void *y, int ld, yolo_stream_t stream)
{
    if (!x || !y || R <= 0 || Cc <= 0 || ld < R) return fail(YOLO_E_ARG, "yolo_transpose_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3(nblk(Cc, 64), nblk(R, 64)), dim3(256), 0, STRM(stream), x, (long)R, (long)Cc, (bf16_t *)y, (long)ld);
    return check_launch("yolo_transpose_f32_to_bf16");
}

YOLO_API int yolo_transpose_bf16(const void *x, int R, int Cc, int ldx, void *y, int ldy, yolo_stream_t stream)
{
    if (!x || !y || R <= 0 || Cc <= 0 || ldx < Cc || ldy < R) return fail(YOLO_E_ARG, "yolo_transpose_bf16: bad argument");
    hipLaunchKernelGGL(transpose_bf16_ld_kernel, dim3(nblk(Cc, 64), nblk(R, 64)), dim3(256), 0, STRM(stream), (const bf16_t *)x, R, Cc, ldx, (bf16_t *)y, ldy);
    return check_launch("yolo_transpose_bf16");
}

YOLO_API int yolo_fc_dgrad_to_nhwc(const float *dxT, int N, int C, int H, int W, int halo, const void *y_act, float slope, void *g, yolo_stream_t stream)
{
    if (!dxT || !g || N <= 0 || C <= 0 || H <= 0 || W <= 0 || halo < 0) return fail(YOLO_E_ARG, "yolo_fc_dgrad_to_nhwc: bad argument");
    if ((long)H * W > 65535 || nblk(N, 64) > 65535) return fail(YOLO_E_UNSUPPORTED, "yolo_fc_dgrad_to_nhwc: H*W=%ld or N=%d too large", (long)H * W, N);
    hipLaunchKernelGGL(fc_dgrad_to_nhwc_kernel, dim3(nblk(C, 64), H * W, nblk(N, 64)), dim3(256), 0, STRM(stream), dxT, N, C, H, W, halo, (const bf16_t *)y_act,
                       slope, (bf16_t *)g);
    return check_launch("yolo_fc_dgrad_to_nhwc");
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

YOLO_API int yolo_bias_lrelu_rows_slabs(const float *x, int slabs, const float *bias, int R, int Cc, float slope, void *yb, float *yf, yolo_stream_t stream)
{
    if (!x || (!yb && !yf) || R <= 0 || Cc <= 0 || slabs < 1) return fail(YOLO_E_ARG, "yolo_bias_lrelu_rows: bad argument");
    hipLaunchKernelGGL(bias_lrelu_rows_kernel, dim3(nblk((long)R * Cc, 256)), dim3(256), 0, STRM(stream), x, slabs, bias, R, Cc, slope, (bf16_t *)yb, yf);
    return check_launch("yolo_bias_lrelu_rows");
}

YOLO_API int yolo_bias_lrelu_rows(const float *x, const float *bias, int R, int Cc, float slope, void *yb, float *yf, yolo_stream_t stream)
{
    return yolo_bias_lrelu_rows_slabs(x, 1, bias, R, Cc, slope, yb, yf, stream);
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

YOLO_API int yolo_pack_fc_weight_blocked_hwc(const float *w, int O, int C, int HW, void *wb, yolo_stream_t stream)
{
    if (!w || !wb || O <= 0 || C <= 0 || HW <= 0 || (C & 7) || (((long)C * HW) & 63))
        return fail(YOLO_E_ARG, "yolo_pack_fc_weight_blocked_hwc: bad argument (C must be a multiple of 8, C * HW of 64)");
    const long total8 = (long)((O + 127) / 128) * ((long)C * HW / 64) * 128 * 8;
    hipLaunchKernelGGL(pack_fc_blocked_hwc_kernel, dim3(nblk(total8, 256)), dim3(256), 0, STRM(stream), w, O, C, HW, (bf16_t *)wb);
    return check_launch("yolo_pack_fc_weight_blocked_hwc");
}

YOLO_API int yolo_pack_fc_weight_blocked(const float *w, int O, long K, void *wb, yolo_stream_t stream)
{
    if (!w || !wb || O <= 0 || K <= 0 || (K & 63)) return fail(YOLO_E_ARG, "yolo_pack_fc_weight_blocked: bad argument (K must be a multiple of 64)");
    const long total8 = (long)((O + 127) / 128) * (K / 64) * 128 * 8;
    hipLaunchKernelGGL(pack_fc_blocked_kernel, dim3(nblk(total8, 256)), dim3(256), 0, STRM(stream), w, O, K, (bf16_t *)wb);
    return check_launch("yolo_pack_fc_weight_blocked");
}
