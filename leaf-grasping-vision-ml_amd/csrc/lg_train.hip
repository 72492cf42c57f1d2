// GraspPointCNN training step on gfx950: forward in train mode (batch-statistics BatchNorm, Dropout2d / Dropout with
// explicit keep masks), BCEWithLogits(pos_weight) loss, backward, global-norm gradient clipping and Adam with L2 weight
// decay -- one call = one iteration of the inner loop of scripts/train_model.py:247-265 on the model of
// scripts/utils/ml_grasp_optimizer/model.py:5-128 (every attention type -- 'spatial' is the script's default --,
// any encoder_filters of the reference's sweep).
//
// Convolutions (forward, backward-data, backward-weights) run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation) as implicit GEMMs whose operands are gathered per lane from an LDS-staged halo tile:
//   forward / backward-data : M = output channel, N = pixel, K = (tap, input channel)
//   backward-weights        : M = input channel,  N = output channel, K = pixel, one accumulator per tap; split over
//                             pixel tiles into partial sums that a second kernel adds in a fixed order (deterministic).
// The pixel index runs over (sample, y, x), so small feature maps (8x8, 4x4) fill a tile with several samples.
// Everything else (BN statistics / apply / backward, pooling, attention, classifier, loss, optimizer) is HBM- or
// latency-bound elementwise / reduction work; all reductions are ordered, two runs give identical bits.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/leafgrasp.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float lgt_zero_pad[4];  // zero-initialised source for padded / out-of-range direct-to-LDS lanes

namespace {

// ------------------------------------------------------------------------------------------------ tiles
template <int WI, int TILE>
struct Tile {
    static constexpr int HW = WI * WI;
    static constexpr int TS = HW >= TILE ? 1 : TILE / HW;   // samples per tile
    static constexpr int TR = HW >= TILE ? TILE / WI : WI;  // image rows per tile
    static constexpr int BANDS = WI / TR;
    static constexpr int TW = WI + 2, TH = TR + 2;          // staged rows / columns (halo 1)
    static constexpr int PLANE = TS * TH * TW;              // staged floats per channel
    __device__ static constexpr int halo(int q) {           // offset of the 3x3 window's top-left for tile pixel q
        const int ts = q / (TR * WI), row = (q / WI) % TR, x = q % WI;
        return (ts * TH + row) * TW + x;
    }
};

__device__ inline float block_sum(float v, float* s_red) {   // 256 threads; result in every thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Workgroups of 64 columns x kRG row groups: a row group walks rows rg, rg + kRG, ...; the kRG partial sums of a column
// are added in index order by every thread of the column.
constexpr int kRG = 16;
__device__ inline float rg_sum(float v, float (*s)[64], int f, int rg) {
    __syncthreads();
    s[rg][f] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < kRG; k++) t += s[k][f];
    return t;
}

// ------------------------------------------------------------------------------------------------ conv fwd / dgrad
// out[n][co][y][x] = bias[co] + sum_{tap,ci} wp[tap][ci][co] * in[n][ci][y+ky-1][x+kx-1]     (zero padding)
// Workgroup = 4 waves = WPX x WCO x KS waves over (pixels, output channels, slices of the staged input channels); a wave
// owns PB x CB blocks of 32 pixels x 32 channels.  Three shapes:
//   <PB 2, CB 2, WPX 4, KS 1>  256 pixels x 64 channels, 4 MFMAs per 4 LDS operand reads   -- large batches
//   <PB 1, CB 1, WPX 2, KS 1>   64 pixels x 64 channels, 4x the workgroups                  -- small batches
//   <PB 1, CB 1, WPX 1, KS 4>   32 pixels x 32 channels, the four waves split K             -- the reference's batch of 16,
//                                where a 8x8 layer has 1024 pixels in total: 256 workgroups instead of 16
template <int WI, int PB, int CB, int WPX, int KS, int KC>
__global__ __launch_bounds__(256) void lgt_conv_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N,
                                                       int CI, int CO) {
    constexpr int WCO = 4 / (WPX * KS), TILE = 32 * PB * WPX, COT = 32 * CB * WCO, KCT = KC * KS;
    static_assert(WPX * WCO * KS == 4, "4 waves");
    using T = Tile<WI, TILE>;
    constexpr int IN_ELEMS = KCT * T::PLANE, NIN = (IN_ELEMS + 255) / 256, IN_PAD = NIN * 256;
    constexpr int W4_ELEMS = 9 * KCT * (COT / 4), NW4 = (W4_ELEMS + 255) / 256;
    constexpr int BUF = IN_PAD + NW4 * 1024;   // floats per stage
    // one shared object, two stages of [input halo tile | weights], both filled by global_load_lds in load order
    __shared__ __attribute__((aligned(16))) float s_buf[2 * BUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 31, kh = lane >> 5;
    const int wpx = wave % WPX, wco = (wave / WPX) % WCO, ks = wave / (WPX * WCO);
    const int tile = blockIdx.x, n0 = (tile / T::BANDS) * T::TS, y0 = (tile % T::BANDS) * T::TR, co0 = blockIdx.y * COT;

    f32x16 acc[PB][CB];
#pragma unroll
    for (int i = 0; i < PB; i++)
#pragma unroll
        for (int j = 0; j < CB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    int boff[PB];
#pragma unroll
    for (int i = 0; i < PB; i++) boff[i] = T::halo((wpx * PB + i) * 32 + lm);
    const int aoff = wco * CB * 32 + lm;

    // Direct-to-LDS staging: chunk c+1 streams into the other stage while the MFMA loop works on chunk c.  The LDS
    // destination of one wave instruction is a wave-uniform base + lane * size, i.e. slot idx = t + 256 j lands at
    // float idx (input) / float4 idx (weights): the images are linear in idx.  Zero padding of the convolution, channels
    // past CI / CO and unused slots are fetched from a zeroed device word.  Source offsets are chunk-invariant apart from
    // the channel base and computed once.
    int off_in[NIN], ci_in[NIN], off_w[NW4], ci_w[NW4];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int idx = t + 256 * j, ci = idx / T::PLANE, r = idx % T::PLANE;
        const int ts = r / (T::TH * T::TW), ry = (r / T::TW) % T::TH, rx = r % T::TW;
        const int gy = y0 - 1 + ry, gx = rx - 1, n = n0 + ts;
        const bool ok = idx < IN_ELEMS && n < N && gy >= 0 && gy < WI && gx >= 0 && gx < WI;
        off_in[j] = ok ? ((n * CI + ci) * WI + gy) * WI + gx : -1;
        ci_in[j] = ci;
    }
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        const int idx = t + 256 * j, q = idx % (COT / 4), rest = idx / (COT / 4), ci = rest % KCT, tap = rest / KCT;
        off_w[j] = (idx < W4_ELEMS && co0 + 4 * q < CO) ? (tap * CI + ci) * CO + co0 + 4 * q : -1;
        ci_w[j] = ci;
    }
    auto issue_chunk = [&](int c0, int stage) {
        float* sb = s_buf + stage * BUF;
        const float* in_c = in + (size_t)c0 * WI * WI;
        const float* w_c = wp + (size_t)c0 * CO;
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const float* src = (off_in[j] >= 0 && c0 + ci_in[j] < CI) ? in_c + off_in[j] : lgt_zero_pad;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + 256 * j + 64 * wave), 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NW4; j++) {
            const float* src = (off_w[j] >= 0 && c0 + ci_w[j] < CI) ? w_c + off_w[j] : lgt_zero_pad;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + IN_PAD + 4 * (256 * j + 64 * wave)),
                                             16, 0, 0);
        }
    };
    issue_chunk(0, 0);
    int stage = 0;
    for (int c0 = 0; c0 < CI; c0 += KCT, stage ^= 1) {
        __syncthreads();   // own loads landed (vmcnt(0)) + every wave is done with the other stage
        if (c0 + KCT < CI) issue_chunk(c0 + KCT, stage ^ 1);
        const float* s_in = s_buf + stage * BUF;
        const float* s_w = s_in + IN_PAD;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int k0 = 0; k0 < KC; k0 += 2) {
                const int ci = ks * KC + k0 + kh;   // the wave's slice of the staged channels
                float a[CB], b[PB];
#pragma unroll
                for (int j = 0; j < CB; j++) a[j] = s_w[(tap * KCT + ci) * COT + aoff + 32 * j];
#pragma unroll
                for (int i = 0; i < PB; i++) b[i] = s_in[ci * T::PLANE + boff[i] + ky * T::TW + kx];
#pragma unroll
                for (int i = 0; i < PB; i++)
#pragma unroll
                    for (int j = 0; j < CB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i], acc[i][j], 0, 0, 0);
            }
        }
    }
    if (KS > 1) {   // the K slices of a tile are added in slice order by the slice-0 wave
        static_assert(KS == 1 || (PB == 1 && CB == 1 && WPX == 1), "K split: one 32 x 32 tile per workgroup");
        __syncthreads();
        if (ks > 0)
#pragma unroll
            for (int r = 0; r < 16; r++) s_buf[(ks - 1) * 1024 + r * 64 + lane] = acc[0][0][r];
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 0; k < KS - 1; k++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[0][0][r] += s_buf[k * 1024 + r * 64 + lane];
    }
    // epilogue: the 16 bias values of a channel block are fetched together (a load + wait per store would cost as much
    // as the K loop of a 64-channel layer); CO is a multiple of 32, so a channel block is inside or outside as a whole
    size_t pix[PB];
    bool okp[PB];
#pragma unroll
    for (int i = 0; i < PB; i++) {
        const int q = (wpx * PB + i) * 32 + lm;
        const int n = n0 + q / (T::TR * WI), y = y0 + (q / WI) % T::TR, x = q % WI;
        okp[i] = n < N;
        pix[i] = (size_t)n * CO * WI * WI + (size_t)y * WI + x;
    }
#pragma unroll
    for (int j = 0; j < CB; j++) {
        const int cob = co0 + (wco * CB + j) * 32;
        if (cob >= CO) continue;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) bv[r] = bias ? bias[cob + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.0f;
#pragma unroll
        for (int i = 0; i < PB; i++) {
            if (!okp[i]) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = cob + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[pix[i] + (size_t)co * WI * WI] = acc[i][j][r] + bv[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv wgrad
// partial[s][tap][ci][co] = sum over the pixel tiles of slice s of a[n][ci][y+ky-1][x+kx-1] * dx[n][co][y][x]
// Workgroup: 32 ci x 64 co, six waves = 2 blocks of 32 output channels x 3 kernel rows; a wave owns the three tap
// accumulators of its row (M = 32 input channels read at the tap's halo offset, N = 32 output channels, K = pixels).
// The input halo tile and the dX tile of kWgTile pixels stream into a double-buffered LDS image with global_load_lds
// (slot idx = t + 384 j lands at float idx; the odd row strides that keep the operand reads conflict-free are slots that
// fetch the zero word); source offsets relative to the tile origin are computed once.  ~100 VGPRs, 62 KB LDS: two
// workgroups = 12 waves per CU, three per SIMD.
constexpr int kWgTile = 64, kWgThreads = 384;
template <int WI>
__global__ __launch_bounds__(kWgThreads, 2) void lgt_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ dx,
                                                                float* __restrict__ partial, int N, int CI, int CO, int ntiles) {
    using T = Tile<WI, kWgTile>;
    constexpr int SD = kWgTile + 1, SI = T::PLANE | 1, NT = kWgThreads;
    constexpr int NSD = (64 * SD + NT - 1) / NT, NSI = (32 * SI + NT - 1) / NT, BUF = (NSD + NSI) * NT;
    __shared__ __attribute__((aligned(16))) float s_buf[2 * BUF];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lm = lane & 31, kh = lane >> 5;
    const int wc = wave & 1, ky = wave >> 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 32;
    const bool active = co0 + wc * 32 < CO;

    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[k][r] = 0.0f;

    // tile-invariant part of the source offsets (relative to sample n0, row y0) and, packed four to a register, a class byte
    // per slot: bit 0 = never loads (padding slot, channel or column outside), bit 1 = top halo row (outside the image when
    // the tile starts at row 0), bit 2 = bottom halo row (outside when the tile ends at the last row), bits 4-7 = sample
    // within the tile.  Per tile a wave-uniform mask says which classes are dead: two VALU per slot decide the source.
    int rel_d[NSD], rel_i[NSI];
    unsigned cls_i[(NSI + 3) / 4], cls_d[(NSD + 3) / 4];
#pragma unroll
    for (int j = 0; j < (NSI + 3) / 4; j++) cls_i[j] = 0;
#pragma unroll
    for (int j = 0; j < (NSD + 3) / 4; j++) cls_d[j] = 0;
#pragma unroll
    for (int j = 0; j < NSD; j++) {
        const int idx = t + NT * j, co = idx / SD, q = idx % SD;
        const int ts = q / (T::TR * WI), row = (q / WI) % T::TR, x = q % WI;
        const bool ok = idx < 64 * SD && q < kWgTile && co0 + co < CO;
        rel_d[j] = ok ? ((ts * CO + co0 + co) * WI + row) * WI + x : 0;
        cls_d[j >> 2] |= (unsigned)((ok ? 0 : 1) | (ts << 4)) << (8 * (j & 3));
    }
#pragma unroll
    for (int j = 0; j < NSI; j++) {
        const int idx = t + NT * j, ci = idx / SI, r = idx % SI;
        const int ts = r / (T::TH * T::TW), ry = (r / T::TW) % T::TH, rx = r % T::TW;
        const bool ok = idx < 32 * SI && r < T::PLANE && ci0 + ci < CI && rx >= 1 && rx <= WI;
        rel_i[j] = ok ? ((ts * CI + ci0 + ci) * WI + ry - 1) * WI + rx - 1 : 0;
        cls_i[j >> 2] |= (unsigned)((ok ? 0 : 1) | (ry == 0 ? 2 : 0) | (ry == T::TH - 1 ? 4 : 0) | (ts << 4)) << (8 * (j & 3));
    }
    auto issue_tile = [&](int tile, int stage) {
        float* sb = s_buf + stage * BUF;
        const int n0 = (tile / T::BANDS) * T::TS, y0 = (tile % T::BANDS) * T::TR;
        const float* d0 = dx + ((size_t)n0 * CO * WI + y0) * WI;
        const float* a0 = a + ((size_t)n0 * CI * WI + y0) * WI;
        const unsigned dead = 1u | (y0 == 0 ? 2u : 0u) | (y0 + T::TR >= WI ? 4u : 0u);   // wave-uniform
#pragma unroll
        for (int j = 0; j < NSD; j++) {
            bool off = (cls_d[j >> 2] & (1u << (8 * (j & 3)))) != 0;
            if (T::TS > 1) off = off || n0 + (int)((cls_d[j >> 2] >> (8 * (j & 3) + 4)) & 15u) >= N;
            const float* src = off ? lgt_zero_pad : d0 + rel_d[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + NT * j + 64 * wave), 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NSI; j++) {
            bool off = (cls_i[j >> 2] & (dead << (8 * (j & 3)))) != 0;
            if (T::TS > 1) off = off || n0 + (int)((cls_i[j >> 2] >> (8 * (j & 3) + 4)) & 15u) >= N;
            const float* src = off ? lgt_zero_pad : a0 + rel_i[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + NT * (NSD + j) + 64 * wave), 4, 0, 0);
        }
    };
    int stage = 0;
    if ((int)blockIdx.x < ntiles) issue_tile(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, stage ^= 1) {
        __syncthreads();   // own loads landed (vmcnt(0)) + every wave is done with the other stage
        if (tile + (int)gridDim.x < ntiles) issue_tile(tile + gridDim.x, stage ^ 1);
        if (active) {
            // pixel q0 + kh sits right of the even pixel q0 in the same image row: halo(q0 + kh) = halo(q0) + kh, so the
            // fully unrolled loop reads at compile-time offsets from two per-lane base pointers (no VALU between MFMAs)
            const float* bb = s_buf + stage * BUF + (wc * 32 + lm) * SD + kh;
            const float* ab = s_buf + stage * BUF + NSD * NT + lm * SI + ky * T::TW + kh;
#pragma unroll
            for (int q0 = 0; q0 < kWgTile; q0 += 2) {
                const float b = bb[q0];
                const float* ai = ab + T::halo(q0);
#pragma unroll
                for (int kx = 0; kx < 3; kx++) acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[kx], b, acc[kx], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    const int co = co0 + wc * 32 + lm;
#pragma unroll
    for (int kx = 0; kx < 3; kx++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (ci < CI && co < CO) partial[(((size_t)blockIdx.x * 9 + ky * 3 + kx) * CI + ci) * CO + co] = acc[kx][r];
        }
}

// gw[co][ci][tap] = sum_s partial[s][tap][ci][co]   (fixed order)
__global__ void lgt_wreduce_kernel(const float* __restrict__ partial, int S, int CI, int CO, float* __restrict__ gw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, n = 9 * CI * CO;
    if (e >= n) return;
    float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // four interleaved chains (loads in flight), combined in a fixed order
    int k = 0;
    for (; k + 4 <= S; k += 4)
#pragma unroll
        for (int u = 0; u < 4; u++) s4[u] += partial[(size_t)(k + u) * n + e];
    for (; k < S; k++) s4[0] += partial[(size_t)k * n + e];
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
    gw[((size_t)co * CI + ci) * 9 + tap] = s;
}

// w[co][ci][tap] -> wpf[tap][ci][co] (forward) and wpd[8-tap][co][ci] (backward-data: taps mirrored, channels swapped);
// all conv layers in one launch
struct PackTable { const float* w[8]; float* wpf[8]; float* wpd[8]; int ci[8], co[8]; unsigned end[8]; int n; };
__global__ void lgt_pack_kernel(PackTable T) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    int l = 0;
    while (l < T.n && g >= T.end[l]) l++;
    if (l >= T.n) return;
    const int e = (int)(g - (l ? T.end[l - 1] : 0u)), CI = T.ci[l], CO = T.co[l];
    const int tap = e % 9, ci = (e / 9) % CI, co = e / (9 * CI);
    const float v = T.w[l][e];
    T.wpf[l][((size_t)tap * CI + ci) * CO + co] = v;
    T.wpd[l][((size_t)(8 - tap) * CO + co) * CI + ci] = v;
}

// ------------------------------------------------------------------------------------------------ BatchNorm2d
// Workgroup (c, s) reduces channel c over sample chunk s: count, mean and M2 = sum (x - chunk mean)^2 (two passes, the
// second one hits L2).  The chunks are combined in index order with Chan's formula -- by the same workgroup when there
// is one chunk, else by lgt_bn_finish_kernel: mean, biased variance -> rstd; running statistics as nn.BatchNorm2d
// (momentum 0.1, unbiased variance in the running estimate).
__device__ inline void bn_finish(int c, float cnt, float mu, float m2, float eps, float momentum, float* mean, float* rstd,
                                 float* run_mean, float* run_var) {
    const float var = m2 / cnt;
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + eps);
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * var * (cnt / (cnt > 1.0f ? cnt - 1.0f : 1.0f));
}
__global__ __launch_bounds__(256) void lgt_bn_stats_kernel(const float* __restrict__ x, int N, int C, int HW, int chunk,
                                                           float eps, float momentum, float* __restrict__ part,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var) {
    __shared__ float s_red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int na = blockIdx.y * chunk, nb = min(N, na + chunk), M = (nb - na) * HW;
    const float* xc = x + ((size_t)na * C + c) * HW;
    const size_t sn = (size_t)C * HW;
    float s = 0.0f;
#pragma unroll 4
    for (int i = t; i < M; i += 256) s += xc[(size_t)(i / HW) * sn + i % HW];
    const float mu = block_sum(s, s_red) / (float)M;
    float v = 0.0f;
#pragma unroll 4
    for (int i = t; i < M; i += 256) {
        const float d = xc[(size_t)(i / HW) * sn + i % HW] - mu;
        v += d * d;
    }
    const float m2 = block_sum(v, s_red);
    if (t == 0) {
        if (gridDim.y == 1) bn_finish(c, (float)M, mu, m2, eps, momentum, mean, rstd, run_mean, run_var);
        else {
            float* q = part + ((size_t)c * gridDim.y + blockIdx.y) * 3;
            q[0] = (float)M; q[1] = mu; q[2] = m2;
        }
    }
}
__global__ void lgt_bn_finish_kernel(const float* __restrict__ part, int C, int S, float eps, float momentum,
                                     float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                     float* __restrict__ run_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float* q = part + (size_t)c * S * 3;
    float cnt = q[0], mu = q[1], m2 = q[2];
    for (int k = 1; k < S; k++) {
        const float nb = q[3 * k], mb = q[3 * k + 1], d = mb - mu, tot = cnt + nb;
        mu += d * (nb / tot);
        m2 += q[3 * k + 2] + d * d * (cnt * nb / tot);
        cnt = tot;
    }
    bn_finish(c, cnt, mu, m2, eps, momentum, mean, rstd, run_mean, run_var);
}
// out[k] = sum_s part[k][s]   (k < K, fixed order)
__global__ void lgt_rowsum_kernel(const float* __restrict__ part, int K, int S, float* __restrict__ o0, float* __restrict__ o1,
                                  int K0) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.0f;
    for (int j = 0; j < S; j++) s += part[(size_t)k * S + j];
    if (k < K0) o0[k] = s; else o1[k - K0] = s;
}

__device__ inline float bn_relu(float x, float mu, float rs, float g, float b) { return fmaxf((x - mu) * rs * g + b, 0.0f); }

// first maximum of the 2x2 cell in row-major scan order (nn.MaxPool2d routes the gradient there)
__device__ inline int argmax4(const float y[4]) {
    int k = 0;
    float m = y[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (y[i] > m) { m = y[i]; k = i; }
    return k;
}

// out = relu(bn(x))                                   (POOL = false)
// out = maxpool2x2(relu(bn(x))) * dmask[n][c]          (POOL = true: Dropout2d keep mask, already scaled by 1/(1-p))
template <bool POOL>
__global__ void lgt_bn_act_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ gam,
                                  const float* __restrict__ bet, const float* __restrict__ dmask,
                                  float* __restrict__ out, int N, int C, int WI) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!POOL) {
        const int HW = WI * WI;
        if (idx >= (size_t)N * C * HW) return;
        const int c = (idx / HW) % C;
        out[idx] = bn_relu(x[idx], mean[c], rstd[c], gam[c], bet[c]);
    } else {
        const int WO = WI / 2;
        if (idx >= (size_t)N * C * WO * WO) return;
        const int xo = idx % WO, yo = (idx / WO) % WO;
        const size_t nc = idx / (WO * WO);
        const int c = nc % C;
        const float* xp = x + (nc * WI + 2 * yo) * WI + 2 * xo;
        const float mu = mean[c], rs = rstd[c], g = gam[c], b = bet[c];
        const float m = fmaxf(fmaxf(bn_relu(xp[0], mu, rs, g, b), bn_relu(xp[1], mu, rs, g, b)),
                              fmaxf(bn_relu(xp[WI], mu, rs, g, b), bn_relu(xp[WI + 1], mu, rs, g, b)));
        out[idx] = m * dmask[nc];
    }
}

// dgamma[c] = sum dy * xhat, dbeta[c] = sum dy with dy = dout routed back through (dropout, max-pool,) ReLU
template <bool POOL>
__global__ __launch_bounds__(256) void lgt_bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gam, const float* __restrict__ bet,
                                                                const float* __restrict__ dmask, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ part, int chunk,
                                                                float* __restrict__ dx, int N, int C, int WI) {
    __shared__ float s_red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int na = blockIdx.y * chunk, nb = min(N, na + chunk);
    const float mu = mean[c], rs = rstd[c], g = gam[c], b = bet[c];
    float sg = 0.0f, sb = 0.0f;
    if (!POOL) {
        const int HW = WI * WI, M = (nb - na) * HW;
#pragma unroll 4
        for (int i = t; i < M; i += 256) {
            const size_t e = ((size_t)(na + i / HW) * C + c) * HW + i % HW;
            const float xh = (x[e] - mu) * rs;
            const float dy = (xh * g + b > 0.0f) ? dout[e] : 0.0f;
            sg += dy * xh;
            sb += dy;
        }
    } else {
        const int WO = WI / 2, HWO = WO * WO, M = (nb - na) * HWO;
        for (int i = t; i < M; i += 256) {
            const int n = na + i / HWO, r = i % HWO, yo = r / WO, xo = r % WO;
            const size_t nc = (size_t)n * C + c;
            const float* xp = x + (nc * WI + 2 * yo) * WI + 2 * xo;
            const float xh[4] = {(xp[0] - mu) * rs, (xp[1] - mu) * rs, (xp[WI] - mu) * rs, (xp[WI + 1] - mu) * rs};
            const float y[4] = {fmaxf(xh[0] * g + b, 0.0f), fmaxf(xh[1] * g + b, 0.0f), fmaxf(xh[2] * g + b, 0.0f),
                                fmaxf(xh[3] * g + b, 0.0f)};
            const int k = argmax4(y);
            const float dy = y[k] > 0.0f ? dout[nc * HWO + r] * dmask[nc] : 0.0f;
            sg += dy * xh[k];
            sb += dy;
        }
    }
    sg = block_sum(sg, s_red);
    sb = block_sum(sb, s_red);
    if (t == 0) {
        if (gridDim.y == 1) { dgamma[c] = sg; dbeta[c] = sb; }
        else {   // part[0..C) rows = dgamma chunks, part[C..2C) rows = dbeta chunks; summed by lgt_rowsum_kernel
            part[(size_t)c * gridDim.y + blockIdx.y] = sg;
            part[(size_t)(C + c) * gridDim.y + blockIdx.y] = sb;
        }
    }
    if (dx == nullptr || gridDim.y != 1) return;
    // small batches: the same workgroup writes dx of its channel (what lgt_bn_bwd_dx_kernel does for large ones)
    const float invM = 1.0f / (float)(N * WI * WI), mb = sb * invM, mg = sg * invM;
    if (!POOL) {
        const int HW = WI * WI, M = N * HW;
#pragma unroll 4
        for (int i = t; i < M; i += 256) {
            const size_t e = ((size_t)(i / HW) * C + c) * HW + i % HW;
            const float xh = (x[e] - mu) * rs;
            const float dy = (xh * g + b > 0.0f) ? dout[e] : 0.0f;
            dx[e] = g * rs * (dy - mb - xh * mg);
        }
    } else {
        const int WO = WI / 2, HWO = WO * WO, M = N * HWO;
        for (int i = t; i < M; i += 256) {
            const int n = i / HWO, r = i % HWO, yo = r / WO, xo = r % WO;
            const size_t nc = (size_t)n * C + c, e0 = (nc * WI + 2 * yo) * WI + 2 * xo;
            const float xh[4] = {(x[e0] - mu) * rs, (x[e0 + 1] - mu) * rs, (x[e0 + WI] - mu) * rs, (x[e0 + WI + 1] - mu) * rs};
            const float y[4] = {fmaxf(xh[0] * g + b, 0.0f), fmaxf(xh[1] * g + b, 0.0f), fmaxf(xh[2] * g + b, 0.0f),
                                fmaxf(xh[3] * g + b, 0.0f)};
            const int k = argmax4(y);
            const float dyk = y[k] > 0.0f ? dout[nc * HWO + r] * dmask[nc] : 0.0f;
            const size_t eo[4] = {e0, e0 + 1, e0 + WI, e0 + WI + 1};
#pragma unroll
            for (int q = 0; q < 4; q++) dx[eo[q]] = g * rs * ((q == k ? dyk : 0.0f) - mb - xh[q] * mg);
        }
    }
}

// dx = gamma * rstd * (dy - dbeta / M - xhat * dgamma / M)
template <bool POOL>
__global__ void lgt_bn_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ gam, const float* __restrict__ bet,
                                     const float* __restrict__ dmask, const float* __restrict__ dgamma,
                                     const float* __restrict__ dbeta, float* __restrict__ dx, int N, int C, int WI) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float invM = 1.0f / (float)(N * WI * WI);
    if (!POOL) {
        const int HW = WI * WI;
        if (idx >= (size_t)N * C * HW) return;
        const int c = (idx / HW) % C;
        const float rs = rstd[c], g = gam[c];
        const float xh = (x[idx] - mean[c]) * rs;
        const float dy = (xh * g + bet[c] > 0.0f) ? dout[idx] : 0.0f;
        dx[idx] = g * rs * (dy - dbeta[c] * invM - xh * dgamma[c] * invM);
    } else {
        const int WO = WI / 2;
        if (idx >= (size_t)N * C * WO * WO) return;
        const int xo = idx % WO, yo = (idx / WO) % WO;
        const size_t nc = idx / (WO * WO);
        const int c = nc % C;
        const size_t e0 = (nc * WI + 2 * yo) * WI + 2 * xo;
        const float mu = mean[c], rs = rstd[c], g = gam[c], b = bet[c];
        const float xh[4] = {(x[e0] - mu) * rs, (x[e0 + 1] - mu) * rs, (x[e0 + WI] - mu) * rs, (x[e0 + WI + 1] - mu) * rs};
        const float y[4] = {fmaxf(xh[0] * g + b, 0.0f), fmaxf(xh[1] * g + b, 0.0f), fmaxf(xh[2] * g + b, 0.0f),
                            fmaxf(xh[3] * g + b, 0.0f)};
        const int k = argmax4(y);
        const float dyk = y[k] > 0.0f ? dout[idx] * dmask[nc] : 0.0f;
        const float mb = dbeta[c] * invM, mg = dgamma[c] * invM;
        const size_t eo[4] = {e0, e0 + 1, e0 + WI, e0 + WI + 1};
#pragma unroll
        for (int i = 0; i < 4; i++) dx[eo[i]] = g * rs * ((i == k ? dyk : 0.0f) - mb - xh[i] * mg);
    }
}

// ------------------------------------------------------------------------------------------------ head
// spatial attention + global average pool: a[n][p] = sigmoid(ba + sum_c wa[c] h[n][c][p]); g[n][c] = mean_p h a
__global__ __launch_bounds__(256) void lgt_att_fwd_kernel(const float* __restrict__ h, const float* __restrict__ wa,
                                                          const float* __restrict__ ba, int spatial, int F, int P,
                                                          float* __restrict__ a, float* __restrict__ g) {
    __shared__ float s_a[64], s_p[256];
    const int n = blockIdx.x, t = threadIdx.x;
    const float* hn = h + (size_t)n * F * P;
    const int p = t % P, cg = t / P, G = 256 / P;     // P divides 256 (1, 4, 16, 64)
    float s = 0.0f;
    if (spatial)
        for (int c = cg; c < F; c += G) s += wa[c] * hn[c * P + p];
    s_p[t] = s;
    __syncthreads();
    if (t < P) {
        float v = 1.0f;
        if (spatial) {
            float z = ba[0];
            for (int k = 0; k < G; k++) z += s_p[k * P + t];
            v = 1.0f / (1.0f + expf(-z));
        }
        s_a[t] = v;
        a[(size_t)n * P + t] = v;
    }
    __syncthreads();
    for (int c = t; c < F; c += 256) {
        float z = 0.0f;
        for (int q = 0; q < P; q++) z += hn[c * P + q] * s_a[q];
        g[(size_t)n * F + c] = z / (float)P;
    }
}

// dh[n][c][p] = dg[n][c] a[p] / P + ds[p] wa[c],  ds[p] = a (1-a) sum_c dg[c] h[c][p] / P;
// part[n][c] = sum_p ds[p] h[c][p] (-> d wa),  part[n][F] = sum_p ds[p] (-> d ba)
__global__ __launch_bounds__(256) void lgt_att_bwd_kernel(const float* __restrict__ h, const float* __restrict__ a,
                                                          const float* __restrict__ dg, const float* __restrict__ wa,
                                                          int spatial, int F, int P, float* __restrict__ dh,
                                                          float* __restrict__ part, const float* __restrict__ dm) {
    __shared__ float s_a[64], s_ds[64], s_p[256];
    const int n = blockIdx.x, t = threadIdx.x;
    const float* hn = h + (size_t)n * F * P;
    const float* dgn = dg + (size_t)n * F;
    const int p = t % P, cg = t / P, G = 256 / P;
    float s = 0.0f;
    if (spatial)
        for (int c = cg; c < F; c += G) s += dgn[c] * hn[c * P + p];
    s_p[t] = s;
    __syncthreads();
    if (t < P) {
        const float av = a[(size_t)n * P + t];
        float z = 0.0f;
        for (int k = 0; k < G; k++) z += s_p[k * P + t];
        s_a[t] = av;
        s_ds[t] = spatial ? z / (float)P * av * (1.0f - av) : 0.0f;
    }
    __syncthreads();
    for (int c = t; c < F; c += 256) {
        const float d = dgn[c] / (float)P, w = spatial ? wa[c] : 0.0f;
        const float dmc = dm ? dm[(size_t)n * F + c] / (float)P : 0.0f;   // channel attention's path through mean_p h
        float pw = 0.0f;
        for (int q = 0; q < P; q++) {
            dh[((size_t)n * F + c) * P + q] = d * s_a[q] + s_ds[q] * w + dmc;
            pw += s_ds[q] * hn[c * P + q];
        }
        if (spatial) part[(size_t)n * (F + 1) + c] = pw;
    }
    if (spatial && t == 0) {
        float z = 0.0f;
        for (int q = 0; q < P; q++) z += s_ds[q];
        part[(size_t)n * (F + 1) + F] = z;
    }
}

__global__ __launch_bounds__(64 * kRG) void lgt_colsum_kernel(const float* __restrict__ part, int N, int K,
                                                              float* __restrict__ out) {
    __shared__ float s_p[kRG][64];
    const int f = threadIdx.x, rg = threadIdx.y, k = min((int)blockIdx.x * 64 + f, K - 1);
    float s = 0.0f;
    for (int n = rg; n < N; n += kRG) s += part[(size_t)n * K + k];
    s = rg_sum(s, s_p, f, rg);
    if (rg == 0 && (int)blockIdx.x * 64 + f < K) out[k] = s;
}

// Channel attention (model.py:37-44, 52-58): ca = sigmoid(W2 relu(W1 m + b1) + b2) with m = mean_p h; pooled feature
// g2 = ca * gs, gs = the (spatially attended, for 'hybrid') mean.  The two 1x1 convolutions run as the classifier's Linear
// kernels on [N][F] / [N][F/16]; these are the elementwise pieces.
__global__ void lgt_relu_kernel(const float* __restrict__ z, float* __restrict__ r, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = fmaxf(z[i], 0.0f);
}
__global__ void lgt_catt_gate_kernel(const float* __restrict__ z2, const float* __restrict__ gs, float* __restrict__ ca,
                                     float* __restrict__ g2, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = 1.0f / (1.0f + expf(-z2[i]));
    ca[i] = c;
    g2[i] = c * gs[i];
}
// dgs = dg2 * ca, dz2 = dg2 * gs * ca (1 - ca)
__global__ void lgt_catt_gate_bwd_kernel(const float* __restrict__ dg2, const float* __restrict__ ca,
                                         const float* __restrict__ gs, float* __restrict__ dgs, float* __restrict__ dz2,
                                         size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = ca[i], d = dg2[i];
    dgs[i] = d * c;
    dz2[i] = d * gs[i] * c * (1.0f - c);
}
// dz1 = dr * (z1 > 0)
__global__ void lgt_relu_bwd_kernel(const float* __restrict__ dr, const float* __restrict__ z1, float* __restrict__ dz1,
                                    size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz1[i] = z1[i] > 0.0f ? dr[i] : 0.0f;
}

// Y[n][o] = b[o] + sum_i X[n][i] W[o][i]; one wave per (o, kFcRows samples), the weight row held in registers (I <= 1024).
// Four samples per wave: one dependent round of loads + shuffles per launch (16 per wave cost 18 us at batch 16).
constexpr int kFcRows = 4;
__global__ __launch_bounds__(256) void lgt_fc_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ Y, int N, int I,
                                                         int O) {
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = lane + 64 * k < I ? W[(size_t)o * I + lane + 64 * k] : 0.0f;
    const float bo = b[o];
    const int nk = (I + 63) / 64;
    const int nb = blockIdx.y * kFcRows, ne = min(N, nb + kFcRows);
    for (int n = nb; n < ne; n += 4) {   // four independent samples at a time: their loads and shuffle chains overlap
        float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nk && n + u < ne) s4[u] += (lane + 64 * k < I ? X[(size_t)(n + u) * I + lane + 64 * k] : 0.0f) * w[k];
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) s4[u] += __shfl_xor(s4[u], k, 64);
            if (lane == 0 && n + u < ne) Y[(size_t)(n + u) * O + o] = s4[u] + bo;
        }
    }
}

// BatchNorm1d (batch statistics) + ReLU + Dropout keep mask.  Workgroup = 64 features x kRG row groups over the samples.
__global__ __launch_bounds__(64 * kRG) void lgt_bn1d_fwd_kernel(const float* __restrict__ U, const float* __restrict__ gam,
                                    const float* __restrict__ bet, const float* __restrict__ mask, float* __restrict__ Y,
                                    float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                    float* __restrict__ run_var, int N, int O, float eps, float momentum) {
    __shared__ float s_p[kRG][64];
    const int f = threadIdx.x, rg = threadIdx.y, o = min((int)blockIdx.x * 64 + f, O - 1);
    const bool own = (int)blockIdx.x * 64 + f < O;
    float s = 0.0f;
    for (int n = rg; n < N; n += kRG) s += U[(size_t)n * O + o];
    const float mu = rg_sum(s, s_p, f, rg) / (float)N;
    float v = 0.0f;
    for (int n = rg; n < N; n += kRG) {
        const float d = U[(size_t)n * O + o] - mu;
        v += d * d;
    }
    const float var = rg_sum(v, s_p, f, rg) / (float)N, rs = 1.0f / sqrtf(var + eps);
    if (!own) return;
    if (rg == 0) {
        mean[o] = mu;
        rstd[o] = rs;
        run_mean[o] = (1.0f - momentum) * run_mean[o] + momentum * mu;
        run_var[o] = (1.0f - momentum) * run_var[o] + momentum * var * ((float)N / (float)(N > 1 ? N - 1 : 1));
    }
    const float g = gam[o], b = bet[o];
    for (int n = rg; n < N; n += kRG)
        Y[(size_t)n * O + o] = fmaxf((U[(size_t)n * O + o] - mu) * rs * g + b, 0.0f) * mask[(size_t)n * O + o];
}

__global__ __launch_bounds__(64 * kRG) void lgt_bn1d_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ U,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gam, const float* __restrict__ bet,
                                    const float* __restrict__ mask, float* __restrict__ dU, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int N, int O) {
    __shared__ float s_p[kRG][64];
    const int f = threadIdx.x, rg = threadIdx.y, o = min((int)blockIdx.x * 64 + f, O - 1);
    const bool own = (int)blockIdx.x * 64 + f < O;
    const float mu = mean[o], rs = rstd[o], g = gam[o], b = bet[o];
    float sg = 0.0f, sb = 0.0f;
    for (int n = rg; n < N; n += kRG) {
        const float xh = (U[(size_t)n * O + o] - mu) * rs;
        const float dy = (xh * g + b > 0.0f) ? dY[(size_t)n * O + o] * mask[(size_t)n * O + o] : 0.0f;
        sg += dy * xh;
        sb += dy;
    }
    sg = rg_sum(sg, s_p, f, rg);
    sb = rg_sum(sb, s_p, f, rg);
    if (!own) return;
    if (rg == 0) { dgamma[o] = sg; dbeta[o] = sb; }
    const float invN = 1.0f / (float)N;
    for (int n = rg; n < N; n += kRG) {
        const float xh = (U[(size_t)n * O + o] - mu) * rs;
        const float dy = (xh * g + b > 0.0f) ? dY[(size_t)n * O + o] * mask[(size_t)n * O + o] : 0.0f;
        dU[(size_t)n * O + o] = g * rs * (dy - sb * invN - xh * sg * invN);
    }
}

// dX[n][i] = sum_o dY[n][o] W[o][i]
__global__ __launch_bounds__(64 * kRG) void lgt_fc_bwd_x_kernel(const float* __restrict__ dY, const float* __restrict__ W,
                                                                float* __restrict__ dX, int N, int I, int O) {
    __shared__ float s_p[kRG][64];
    const int f = threadIdx.x, rg = threadIdx.y, i = min((int)blockIdx.x * 64 + f, I - 1), n = blockIdx.y;
    float s = 0.0f;
    for (int o = rg; o < O; o += kRG) s += dY[(size_t)n * O + o] * W[(size_t)o * I + i];
    s = rg_sum(s, s_p, f, rg);
    if (rg == 0 && (int)blockIdx.x * 64 + f < I) dX[(size_t)n * I + i] = s;
}

// dW[o][i] = sum_n dY[n][o] X[n][i], db[o] = sum_n dY[n][o]
__global__ __launch_bounds__(64 * kRG) void lgt_fc_bwd_w_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                float* __restrict__ dW, float* __restrict__ db, int N, int I,
                                                                int O) {
    __shared__ float s_p[kRG][64];
    const int f = threadIdx.x, rg = threadIdx.y, i = min((int)blockIdx.x * 64 + f, I - 1), o = blockIdx.y;
    float s = 0.0f, sb = 0.0f;
    for (int n = rg; n < N; n += kRG) {
        const float d = dY[(size_t)n * O + o];
        s += d * X[(size_t)n * I + i];
        sb += d;
    }
    s = rg_sum(s, s_p, f, rg);
    sb = rg_sum(sb, s_p, f, rg);
    if (rg == 0 && (int)blockIdx.x * 64 + f < I) dW[(size_t)o * I + i] = s;
    if (rg == 0 && f == 0 && blockIdx.x == 0) db[o] = sb;
}

// hyper-parameters of a step, refreshed on the device before every step (outside the captured graph)
struct AdamHp { float lr, beta1, beta2, eps, weight_decay, max_norm, pos_weight, pad; };
// BCEWithLogitsLoss(pos_weight), mean over the batch (train_model.py:221,251): loss and d loss / d logit
__global__ __launch_bounds__(256) void lgt_loss_kernel(const float* __restrict__ z, const float* __restrict__ y, int N,
                                                       const AdamHp* __restrict__ hp, float* __restrict__ loss,
                                                       float* __restrict__ dz) {
    __shared__ float s_red[4];
    const float pos_weight = hp->pos_weight;
    float s = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float zn = z[n], yn = y[n];
        // softplus(-z) = max(-z, 0) + log1p(exp(-|z|))
        const float sp_neg = fmaxf(-zn, 0.0f) + log1pf(expf(-fabsf(zn)));
        const float lw = 1.0f + (pos_weight - 1.0f) * yn;
        s += (1.0f - yn) * zn + lw * sp_neg;
        const float sig = 1.0f / (1.0f + expf(-zn));
        dz[n] = ((1.0f - yn) - lw * (1.0f - sig)) / (float)N;
    }
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) loss[0] = s / (float)N;
}

// ------------------------------------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(256) void lgt_sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float s_red[4];
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}


// state[0] = step count (float), state[1] = total gradient norm of this step (before clipping).  One wave adds the
// partial sums in index order and advances the step count when the optimizer runs.
__global__ __launch_bounds__(64) void lgt_norm_kernel(const float* __restrict__ partial, int nparts, int advance,
                                                      float* __restrict__ state) {
    float ss = 0.0f;   // one wave: lane l adds partials l, l + 64, ... in order, then a fixed butterfly
    for (int k = threadIdx.x; k < nparts; k += 64) ss += partial[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (threadIdx.x == 0) {
        state[1] = sqrtf(ss);
        if (advance) state[0] += 1.0f;
    }
}
// clip_grad_norm_(max_norm) + torch.optim.Adam(weight_decay = L2 added to the gradient)
__global__ __launch_bounds__(256) void lgt_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, size_t n,
                                                       const AdamHp* __restrict__ hp_, const float* __restrict__ state) {
    const AdamHp hp = *hp_;
    const float norm = state[1], step = state[0];
    const float coef = hp.max_norm > 0.0f ? fminf(hp.max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
    const float bc1 = 1.0f - powf(hp.beta1, step), bc2 = 1.0f - powf(hp.beta2, step);
    const float step_size = hp.lr / bc1, bc2s = sqrtf(bc2);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float gi = g[i] * coef + hp.weight_decay * p[i];
        const float mi = hp.beta1 * m[i] + (1.0f - hp.beta1) * gi;
        const float vi = hp.beta2 * v[i] + (1.0f - hp.beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + hp.eps));
    }
}

// keep masks of all dropout layers in one launch: 0 or 1/(1-p); counter-based hash of (seed, index)
struct MaskTable { size_t end[8]; float p[8]; int n; };
__global__ void lgt_mask_kernel(float* __restrict__ mask, MaskTable T, const uint64_t* __restrict__ seed_,
                                const float* __restrict__ state) {
    const uint64_t seed = seed_[0] * 0x100000001B3ull + 0x51ull * (uint64_t)state[0];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int l = 0;
    while (l < T.n && i >= T.end[l]) l++;
    if (l >= T.n) return;
    uint64_t zz = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
    zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
    zz ^= zz >> 31;
    const float u = (float)(zz >> 40) * (1.0f / 16777216.0f), p = T.p[l];
    mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace

// ================================================================================================= host side
struct TrainLayer {
    int ci = 0, co = 0, wi = 0;
    bool pool = false;
    size_t w = 0, b = 0, g = 0, be = 0;   // offsets into the flat parameter vector
    size_t rm = 0, rv = 0;                // offsets into the flat buffer vector (running mean / var)
    size_t st = 0;                        // offset into the saved mean / rstd vector
    float *x = nullptr, *out = nullptr;   // conv output (pre-BN), layer output (post BN/ReLU[/pool/dropout])
    float *wpf = nullptr, *wpd = nullptr;
    size_t mask = 0;                      // offset of the block's Dropout2d mask inside one sample's mask row (pool layers)
};
struct TrainFc { int in = 0, out = 0; size_t w = 0, b = 0, g = 0, be = 0, rm = 0, rv = 0, st = 0, mask = 0; float *u = nullptr, *y = nullptr; };

struct lg_trainer {
    int device = 0, n_blocks = 3, att = LG_ATT_SPATIAL, capN = 0, F = 256, P = 16;
    int filters[4] = {64, 128, 256, 0};
    std::vector<TrainLayer> layers;
    TrainFc fc[4];
    size_t att_w = 0, att_b = 0, ca_w1 = 0, ca_b1 = 0, ca_w2 = 0, ca_b2 = 0;
    int hid = 0;                                  // channel attention bottleneck: F / 16
    float *ca_m = nullptr, *ca_z1 = nullptr, *ca_r = nullptr, *ca_z2 = nullptr, *ca_ca = nullptr, *gap2 = nullptr;
    float *ca_dgs = nullptr, *ca_dz2 = nullptr, *ca_dr = nullptr, *ca_dz1 = nullptr, *ca_dm = nullptr, *ca_a1 = nullptr;
    size_t n_params = 0, n_buffers = 0, n_stats = 0, mask_row = 0;
    float drop2d_p = 0.3f, drop_p[3] = {0.5f, 0.5f, 0.4f};
    float *P_ = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *B = nullptr, *mean = nullptr, *rstd = nullptr;
    float *xin = nullptr, *labels = nullptr, *masks = nullptr;
    float *dA[2] = {nullptr, nullptr}, *dX[2] = {nullptr, nullptr}, *partial = nullptr, *bn_part = nullptr;
    size_t partial_floats = 0;
    float *att_a = nullptr, *gap = nullptr, *att_part = nullptr, *dfc[2] = {nullptr, nullptr}, *logits = nullptr, *dz = nullptr;
    float *loss = nullptr, *state = nullptr, *norm_part = nullptr;
    AdamHp* hp = nullptr;
    hipStream_t stream = nullptr, stream_w = nullptr;   // main chain; backward-weights branch
    hipEvent_t ev_dx[8] = {}, ev_wg[8] = {};
    uint64_t* seed_dev = nullptr;
    bool side_stream = false;                    // backward-weights on stream_w (LG_TRAIN_STREAMS=2); default: in line
    int conv_small_below = 256, conv_split_below = 256;   // convolution shape thresholds (launch_conv)
    bool use_graph = false;                      // LG_TRAIN_GRAPH=1: replay the step as a captured graph (measured: no gain)
    std::map<uint64_t, hipGraphExec_t> graphs;   // key: N, masks drawn on the device, optimizer applied
    int64_t steps = 0;
    std::string err;
    std::vector<void*> allocs;
};

namespace {

constexpr int kNormParts = 256;
constexpr int kMaxSplit = 512;                  // backward-weights pixel slices (partial sums) per layer
constexpr size_t kMaxPartialFloats = 96u << 20; // ... bounded by 384 MiB of partial sums
constexpr int kBnSplit = 64;                    // BatchNorm reduction: sample chunks per channel

// pixel slices of a backward-weights launch: enough workgroups (slices x channel blocks) for two per CU, bounded memory
int wgrad_split(int ntiles, size_t nw, int blocks) {
    const size_t cap = std::max<size_t>(8, kMaxPartialFloats / nw);
    const size_t want = std::max<size_t>(8, (size_t)(1024 / std::max(1, blocks)));
    return (int)std::min<size_t>(std::min<size_t>((size_t)ntiles, std::min<size_t>(want, (size_t)kMaxSplit)), cap);
}
// sample chunks of the BatchNorm reductions: about 4096 elements per workgroup, at most 2048 workgroups, never more
// chunks than samples (batch 16 at 32x32: 4 chunks instead of one workgroup walking 16384 elements three times)
int bn_chunks(int N, int C, int HW) {
    const long long M = (long long)N * HW;
    const int want = (int)std::min<long long>(M / 4096, (long long)kBnSplit);
    return std::max(1, std::min(std::min(want, N), std::max(1, 2048 / C)));
}

#define TR_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            tr->err = std::string(#call) + ": " + hipGetErrorString(e_);                                 \
            return LG_ERR_HIP;                                                                           \
        }                                                                                                \
    } while (0)

int dalloc(lg_trainer* tr, float** p, size_t floats) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(floats, 1) * sizeof(float)) != hipSuccess) {
        tr->err = "hipMalloc failed";
        return LG_ERR_NOMEM;
    }
    tr->allocs.push_back(q);
    *p = (float*)q;
    return LG_OK;
}

// workgroups of a shape below which the next smaller shape is launched: one workgroup per CU is where the larger tile
// stops paying (swept over batches 16..512: 256 / 256 is the best pair, e.g. batch 256 3.30 ms vs 3.47 with 512 / 256)
// (the thresholds live in the trainer: conv_small_below / conv_split_below, LG_TRAIN_CONV_SMALL / LG_TRAIN_CONV_SPLIT for tuning runs)
int conv_tiles(int wi, int N, int tile);
void launch_conv(const lg_trainer* tr, int wi, hipStream_t s, const float* in, const float* wp, const float* bias, float* out,
                 int N, int CI, int CO) {
    const unsigned cb = (unsigned)((CO + 63) / 64), cb32 = (unsigned)((CO + 31) / 32);
    const bool small = conv_tiles(wi, N, 256) * (int)cb < tr->conv_small_below;
    const bool split = conv_tiles(wi, N, 64) * (int)cb < tr->conv_split_below && CI >= 32;
#define LGT_CONV(W)                                                                                                    \
    if (split)                                                                                                         \
        hipLaunchKernelGGL((lgt_conv_kernel<W, 1, 1, 1, 4, 8>), dim3(conv_tiles(W, N, 32), cb32), dim3(256), 0, s, in, wp,  \
                           bias, out, N, CI, CO);                                                                      \
    else if (small)                                                                                                    \
        hipLaunchKernelGGL((lgt_conv_kernel<W, 1, 1, 2, 1, 16>), dim3(conv_tiles(W, N, 64), cb), dim3(256), 0, s, in, wp,   \
                           bias, out, N, CI, CO);                                                                      \
    else                                                                                                               \
        hipLaunchKernelGGL((lgt_conv_kernel<W, 2, 2, 4, 1, 8>), dim3(conv_tiles(W, N, 256), cb), dim3(256), 0, s, in, wp,   \
                           bias, out, N, CI, CO);
    switch (wi) {
        case 32: LGT_CONV(32) break;
        case 16: LGT_CONV(16) break;
        case 8: LGT_CONV(8) break;
        default: LGT_CONV(4) break;
    }
#undef LGT_CONV
}
void launch_wgrad(int wi, dim3 grid, hipStream_t s, const float* a, const float* dx, float* partial, int N, int CI, int CO,
                  int ntiles) {
    switch (wi) {
        case 32: hipLaunchKernelGGL(lgt_wgrad_kernel<32>, grid, dim3(kWgThreads), 0, s, a, dx, partial, N, CI, CO, ntiles); break;
        case 16: hipLaunchKernelGGL(lgt_wgrad_kernel<16>, grid, dim3(kWgThreads), 0, s, a, dx, partial, N, CI, CO, ntiles); break;
        case 8: hipLaunchKernelGGL(lgt_wgrad_kernel<8>, grid, dim3(kWgThreads), 0, s, a, dx, partial, N, CI, CO, ntiles); break;
        default: hipLaunchKernelGGL(lgt_wgrad_kernel<4>, grid, dim3(kWgThreads), 0, s, a, dx, partial, N, CI, CO, ntiles); break;
    }
}
int conv_tiles(int wi, int N, int tile) {   // pixel tiles of `tile` pixels covering N samples
    const int hw = wi * wi;
    return hw >= tile ? N * (hw / tile) : (N + tile / hw - 1) / (tile / hw);
}
inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace

extern "C" {

int lg_train_create(int device, int n_blocks, const int32_t* filters, int attention_type, int max_batch, lg_trainer** out) {
    if (!out) return LG_ERR_INVALID;
    *out = nullptr;
    if (!filters || n_blocks < 1 || n_blocks > 4 || max_batch < 2 || max_batch > 8192) return LG_ERR_INVALID;
    if (attention_type < LG_ATT_SPATIAL || attention_type > LG_ATT_NONE) return LG_ERR_UNSUPPORTED;
    for (int b = 0; b < n_blocks; b++)
        if (filters[b] < 32 || filters[b] % 32 != 0 || filters[b] > 1024) return LG_ERR_INVALID;
    if (filters[n_blocks - 1] % 4 != 0) return LG_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return LG_ERR_HIP;
    lg_trainer* tr = new (std::nothrow) lg_trainer;
    if (!tr) return LG_ERR_NOMEM;
    tr->device = device;
    tr->n_blocks = n_blocks;
    tr->att = attention_type;
    tr->capN = max_batch;
    hipSetDevice(device);

    // parameter order = model.parameters() of the reference module (model.py:12-85)
    size_t po = 0, bo = 0, so = 0, mo = 0;
    int c = 9, wi = 32;
    for (int b = 0; b < n_blocks; b++) {
        tr->filters[b] = filters[b];
        for (int k = 0; k < 2; k++) {
            TrainLayer L;
            L.ci = k == 0 ? c : filters[b];
            L.co = filters[b];
            L.wi = wi;
            L.pool = k == 1;
            L.w = po; po += (size_t)L.co * L.ci * 9;
            L.b = po; po += L.co;
            L.g = po; po += L.co;
            L.be = po; po += L.co;
            L.rm = bo; bo += L.co;
            L.rv = bo; bo += L.co;
            L.st = so; so += L.co;
            if (L.pool) { L.mask = mo; mo += L.co; }
            tr->layers.push_back(L);
        }
        c = filters[b];
        wi /= 2;
    }
    const int F = c;
    tr->F = F;
    tr->P = wi * wi;
    if (filters[n_blocks - 1] % 16 != 0 && (attention_type == LG_ATT_CHANNEL || attention_type == LG_ATT_HYBRID)) {
        delete tr;
        return LG_ERR_INVALID;
    }
    if (attention_type == LG_ATT_SPATIAL || attention_type == LG_ATT_HYBRID) {   // attention.0 / spatial_attention.0
        tr->att_w = po; po += F;
        tr->att_b = po; po += 1;
    }
    if (attention_type == LG_ATT_CHANNEL || attention_type == LG_ATT_HYBRID) {   // attention.{1,3} / channel_attention.{1,3}
        tr->hid = F / 16;
        tr->ca_w1 = po; po += (size_t)tr->hid * F;
        tr->ca_b1 = po; po += tr->hid;
        tr->ca_w2 = po; po += (size_t)F * tr->hid;
        tr->ca_b2 = po; po += F;
    }
    const int fin[4] = {F, F, F / 2, F / 4}, fout[4] = {F, F / 2, F / 4, 1};
    for (int k = 0; k < 4; k++) {
        TrainFc& f = tr->fc[k];
        f.in = fin[k];
        f.out = fout[k];
        f.w = po; po += (size_t)f.in * f.out;
        f.b = po; po += f.out;
        if (k < 3) {
            f.g = po; po += f.out;
            f.be = po; po += f.out;
            f.rm = bo; bo += f.out;
            f.rv = bo; bo += f.out;
            f.st = so; so += f.out;
            f.mask = mo; mo += f.out;
        }
    }
    tr->n_params = po;
    tr->n_buffers = bo;
    tr->n_stats = so;
    tr->mask_row = mo;

    const size_t N = (size_t)max_batch;
    int rc = LG_OK;
    auto A = [&](float** p, size_t n) { if (rc == LG_OK) rc = dalloc(tr, p, n); };
    A(&tr->P_, po); A(&tr->G, po); A(&tr->M, po); A(&tr->V, po); A(&tr->B, bo); A(&tr->mean, so); A(&tr->rstd, so);
    A(&tr->xin, N * 9 * 1024); A(&tr->labels, N); A(&tr->masks, N * mo);
    size_t max_act = 0;
    for (auto& L : tr->layers) {
        const size_t full = N * L.co * L.wi * L.wi;
        A(&L.x, full);
        A(&L.out, L.pool ? full / 4 : full);
        A(&L.wpf, (size_t)9 * L.ci * L.co);
        A(&L.wpd, (size_t)9 * L.ci * L.co);
        max_act = std::max(max_act, std::max(full, N * L.ci * L.wi * L.wi));
        const int tiles = wgrad_split(conv_tiles(L.wi, max_batch, kWgTile), (size_t)9 * L.ci * L.co, (int)(cdiv(L.co, 64) * cdiv(L.ci, 32)));
        tr->partial_floats = std::max(tr->partial_floats, (size_t)tiles * 9 * L.ci * L.co);
    }
    A(&tr->dA[0], max_act); A(&tr->dA[1], max_act); A(&tr->dX[0], max_act); A(&tr->dX[1], max_act);
    A(&tr->partial, tr->partial_floats); A(&tr->bn_part, (size_t)1024 * kBnSplit * 3);
    A(&tr->att_a, N * tr->P); A(&tr->gap, N * F); A(&tr->att_part, N * (F + 1));
    if (tr->hid) {
        A(&tr->ca_m, N * F); A(&tr->ca_z1, N * tr->hid); A(&tr->ca_r, N * tr->hid); A(&tr->ca_z2, N * F); A(&tr->ca_ca, N * F);
        A(&tr->gap2, N * F); A(&tr->ca_dgs, N * F); A(&tr->ca_dz2, N * F); A(&tr->ca_dr, N * tr->hid); A(&tr->ca_dz1, N * tr->hid);
        A(&tr->ca_dm, N * F); A(&tr->ca_a1, N * tr->P);
    }
    A(&tr->dfc[0], N * F); A(&tr->dfc[1], N * F); A(&tr->logits, N); A(&tr->dz, N);
    for (int k = 0; k < 4; k++) { A(&tr->fc[k].u, N * tr->fc[k].out); if (k < 3) A(&tr->fc[k].y, N * tr->fc[k].out); }
    A(&tr->loss, 4); A(&tr->state, 4); A(&tr->norm_part, kNormParts);
    float *hp = nullptr, *sd = nullptr;
    A(&hp, sizeof(AdamHp) / sizeof(float));
    tr->hp = (AdamHp*)hp;
    A(&sd, 2);
    tr->seed_dev = (uint64_t*)sd;
    if (const char* e = getenv("LG_TRAIN_GRAPH")) tr->use_graph = atoi(e) != 0;
    if (const char* e = getenv("LG_TRAIN_CONV_SMALL")) tr->conv_small_below = atoi(e);
    if (const char* e = getenv("LG_TRAIN_CONV_SPLIT")) tr->conv_split_below = atoi(e);
    if (rc == LG_OK && hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking) != hipSuccess) rc = LG_ERR_HIP;
    if (rc == LG_OK && hipStreamCreateWithFlags(&tr->stream_w, hipStreamNonBlocking) != hipSuccess) rc = LG_ERR_HIP;
    for (int k = 0; k < 8 && rc == LG_OK; k++)
        if (hipEventCreateWithFlags(&tr->ev_dx[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&tr->ev_wg[k], hipEventDisableTiming) != hipSuccess) rc = LG_ERR_HIP;
    if (rc != LG_OK) {
        for (void* q : tr->allocs) hipFree(q);
        for (int k = 0; k < 8; k++) {
            if (tr->ev_dx[k]) hipEventDestroy(tr->ev_dx[k]);
            if (tr->ev_wg[k]) hipEventDestroy(tr->ev_wg[k]);
        }
        if (tr->stream) hipStreamDestroy(tr->stream);
        if (tr->stream_w) hipStreamDestroy(tr->stream_w);
        delete tr;
        return rc;
    }
    hipMemset(tr->P_, 0, po * 4); hipMemset(tr->G, 0, po * 4); hipMemset(tr->M, 0, po * 4); hipMemset(tr->V, 0, po * 4);
    hipMemset(tr->B, 0, bo * 4); hipMemset(tr->state, 0, 16); hipMemset(tr->loss, 0, 16);
    hipDeviceSynchronize();
    // Backward-weights runs in line on the one stream by default.  On a second stream (LG_TRAIN_STREAMS=2: stream_w, dX double
    // buffered, event fences) it overlaps the BatchNorm-backward / backward-data chain and gains 2-5 % at batches 64..1024 --
    // but each of the ~12 stream hops of a step costs ~15 us (0.65 vs 0.67 ms at batch 16), and how ROCm maps a process's
    // streams to its few hardware queues depends on how many streams exist when these two are created: for 2 of 9 probed
    // counts (tests/tools/streams_probe.py) the two-stream step was 2.2-2.5x slower.
    if (const char* e = getenv("LG_TRAIN_STREAMS")) tr->side_stream = atoi(e) == 2;
    *out = tr;
    return LG_OK;
}

int lg_train_destroy(lg_trainer* tr) {
    if (!tr) return LG_ERR_INVALID;
    hipSetDevice(tr->device);
    hipDeviceSynchronize();
    for (auto& kv : tr->graphs) hipGraphExecDestroy(kv.second);
    if (tr->stream) hipStreamDestroy(tr->stream);
    if (tr->stream_w) hipStreamDestroy(tr->stream_w);
    for (int k = 0; k < 8; k++) {
        if (tr->ev_dx[k]) hipEventDestroy(tr->ev_dx[k]);
        if (tr->ev_wg[k]) hipEventDestroy(tr->ev_wg[k]);
    }
    for (void* q : tr->allocs) hipFree(q);
    delete tr;
    return LG_OK;
}

const char* lg_train_last_error(lg_trainer* tr) { return tr ? tr->err.c_str() : "null trainer"; }

int lg_train_sizes(lg_trainer* tr, int64_t* n_params, int64_t* n_buffers, int64_t* mask_row) {
    if (!tr) return LG_ERR_INVALID;
    if (n_params) *n_params = (int64_t)tr->n_params;
    if (n_buffers) *n_buffers = (int64_t)tr->n_buffers;
    if (mask_row) *mask_row = (int64_t)tr->mask_row;
    return LG_OK;
}

int lg_train_set_state(lg_trainer* tr, const float* params, const float* buffers, const float* exp_avg,
                       const float* exp_avg_sq, int64_t step) {
    if (!tr || !params || !buffers || step < 0) return LG_ERR_INVALID;
    hipSetDevice(tr->device);
    TR_HIP(hipStreamSynchronize(tr->stream));
    TR_HIP(hipMemcpy(tr->P_, params, tr->n_params * 4, hipMemcpyHostToDevice));
    TR_HIP(hipMemcpy(tr->B, buffers, tr->n_buffers * 4, hipMemcpyHostToDevice));
    if (exp_avg) TR_HIP(hipMemcpy(tr->M, exp_avg, tr->n_params * 4, hipMemcpyHostToDevice));
    else TR_HIP(hipMemset(tr->M, 0, tr->n_params * 4));
    if (exp_avg_sq) TR_HIP(hipMemcpy(tr->V, exp_avg_sq, tr->n_params * 4, hipMemcpyHostToDevice));
    else TR_HIP(hipMemset(tr->V, 0, tr->n_params * 4));
    const float st[4] = {(float)step, 0.f, 0.f, 0.f};
    TR_HIP(hipMemcpy(tr->state, st, 16, hipMemcpyHostToDevice));
    tr->steps = step;
    return LG_OK;
}

int lg_train_get_state(lg_trainer* tr, float* params, float* buffers, float* exp_avg, float* exp_avg_sq, float* grads,
                       int64_t* step) {
    if (!tr) return LG_ERR_INVALID;
    hipSetDevice(tr->device);
    TR_HIP(hipStreamSynchronize(tr->stream));
    if (params) TR_HIP(hipMemcpy(params, tr->P_, tr->n_params * 4, hipMemcpyDeviceToHost));
    if (buffers) TR_HIP(hipMemcpy(buffers, tr->B, tr->n_buffers * 4, hipMemcpyDeviceToHost));
    if (exp_avg) TR_HIP(hipMemcpy(exp_avg, tr->M, tr->n_params * 4, hipMemcpyDeviceToHost));
    if (exp_avg_sq) TR_HIP(hipMemcpy(exp_avg_sq, tr->V, tr->n_params * 4, hipMemcpyDeviceToHost));
    if (grads) TR_HIP(hipMemcpy(grads, tr->G, tr->n_params * 4, hipMemcpyDeviceToHost));
    if (step) *step = tr->steps;
    return LG_OK;
}

// Everything of a step whose launch parameters depend only on (N, draw_masks, apply_update): captured into a graph.
static int enqueue_step(lg_trainer* tr, int N, bool draw_masks, int apply_update) {
    hipStream_t s = tr->stream;
    const int F = tr->F, P = tr->P;
    const float eps = 1e-5f, mom = 0.1f;
    const size_t mrow = tr->mask_row;
    if (draw_masks) {
        MaskTable mt = {};
        for (auto& L : tr->layers)
            if (L.pool) { mt.end[mt.n] = (size_t)N * (L.mask + L.co); mt.p[mt.n++] = tr->drop2d_p; }
        for (int k = 0; k < 3; k++) { mt.end[mt.n] = (size_t)N * (tr->fc[k].mask + tr->fc[k].out); mt.p[mt.n++] = tr->drop_p[k]; }
        hipLaunchKernelGGL(lgt_mask_kernel, dim3(cdiv((size_t)N * mrow, 256)), dim3(256), 0, s, tr->masks, mt,
                           (const uint64_t*)tr->seed_dev, (const float*)tr->state);
    }
    TR_HIP(hipMemsetAsync(tr->G, 0, tr->n_params * 4, s));   // conv biases: exact zero gradient (see DESIGN)
    {
        PackTable pt = {};
        unsigned end = 0;
        for (auto& L : tr->layers) {
            const int k = pt.n++;
            pt.w[k] = tr->P_ + L.w; pt.wpf[k] = L.wpf; pt.wpd[k] = L.wpd; pt.ci[k] = L.ci; pt.co[k] = L.co;
            end += (unsigned)(9 * L.ci * L.co);
            pt.end[k] = end;
        }
        hipLaunchKernelGGL(lgt_pack_kernel, dim3(cdiv(end, 256)), dim3(256), 0, s, pt);
    }

    // ---------------- forward
    const float* a = tr->xin;
    for (auto& L : tr->layers) {
        launch_conv(tr, L.wi, s, a, L.wpf, tr->P_ + L.b, L.x, N, L.ci, L.co);
        const int S = bn_chunks(N, L.co, L.wi * L.wi), chunk = (N + S - 1) / S, Sy = (N + chunk - 1) / chunk;
        hipLaunchKernelGGL(lgt_bn_stats_kernel, dim3(L.co, Sy), dim3(256), 0, s, L.x, N, L.co, L.wi * L.wi, chunk, eps, mom,
                           tr->bn_part, tr->mean + L.st, tr->rstd + L.st, tr->B + L.rm, tr->B + L.rv);
        if (Sy > 1)
            hipLaunchKernelGGL(lgt_bn_finish_kernel, dim3(cdiv(L.co, 64)), dim3(64), 0, s, tr->bn_part, L.co, Sy, eps, mom,
                               tr->mean + L.st, tr->rstd + L.st, tr->B + L.rm, tr->B + L.rv);
        if (L.pool) {
            const size_t n = (size_t)N * L.co * L.wi * L.wi / 4;
            hipLaunchKernelGGL(lgt_bn_act_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, L.x, tr->mean + L.st,
                               tr->rstd + L.st, tr->P_ + L.g, tr->P_ + L.be, tr->masks + (size_t)N * L.mask, L.out, N, L.co, L.wi);
        } else {
            const size_t n = (size_t)N * L.co * L.wi * L.wi;
            hipLaunchKernelGGL(lgt_bn_act_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, L.x, tr->mean + L.st,
                               tr->rstd + L.st, tr->P_ + L.g, tr->P_ + L.be, (const float*)nullptr, L.out, N, L.co, L.wi);
        }
        a = L.out;
    }
    const int spatial = tr->att == LG_ATT_SPATIAL || tr->att == LG_ATT_HYBRID;
    const int hid = tr->hid;
    hipLaunchKernelGGL(lgt_att_fwd_kernel, dim3(N), dim3(256), 0, s, a, tr->P_ + tr->att_w, tr->P_ + tr->att_b, spatial, F, P,
                       tr->att_a, tr->gap);
    const float* pooled = tr->gap;
    if (hid) {
        const float* m = tr->gap;   // 'channel': the plain mean is the pooled feature itself
        if (spatial) {              // 'hybrid': the channel branch pools the unattended map
            hipLaunchKernelGGL(lgt_att_fwd_kernel, dim3(N), dim3(256), 0, s, a, tr->P_ + tr->att_w, tr->P_ + tr->att_b, 0, F, P,
                               tr->ca_a1, tr->ca_m);
            m = tr->ca_m;
        }
        const size_t nh = (size_t)N * hid, nf = (size_t)N * F;
        hipLaunchKernelGGL(lgt_fc_fwd_kernel, dim3(cdiv(hid, 4), cdiv(N, kFcRows)), dim3(256), 0, s, m, tr->P_ + tr->ca_w1,
                           tr->P_ + tr->ca_b1, tr->ca_z1, N, F, hid);
        hipLaunchKernelGGL(lgt_relu_kernel, dim3(cdiv(nh, 256)), dim3(256), 0, s, tr->ca_z1, tr->ca_r, nh);
        hipLaunchKernelGGL(lgt_fc_fwd_kernel, dim3(cdiv(F, 4), cdiv(N, kFcRows)), dim3(256), 0, s, tr->ca_r, tr->P_ + tr->ca_w2,
                           tr->P_ + tr->ca_b2, tr->ca_z2, N, hid, F);
        hipLaunchKernelGGL(lgt_catt_gate_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, s, tr->ca_z2, tr->gap, tr->ca_ca, tr->gap2, nf);
        pooled = tr->gap2;
    }
    const float* fin = pooled;
    for (int k = 0; k < 4; k++) {
        TrainFc& f = tr->fc[k];
        hipLaunchKernelGGL(lgt_fc_fwd_kernel, dim3(cdiv(f.out, 4), cdiv(N, kFcRows)), dim3(256), 0, s, fin, tr->P_ + f.w,
                           tr->P_ + f.b, k < 3 ? f.u : tr->logits, N, f.in, f.out);
        if (k < 3) {
            hipLaunchKernelGGL(lgt_bn1d_fwd_kernel, dim3(cdiv(f.out, 64)), dim3(64, kRG), 0, s, f.u, tr->P_ + f.g, tr->P_ + f.be,
                               tr->masks + (size_t)N * f.mask, f.y, tr->mean + f.st, tr->rstd + f.st, tr->B + f.rm,
                               tr->B + f.rv, N, f.out, eps, mom);
            fin = f.y;
        }
    }
    hipLaunchKernelGGL(lgt_loss_kernel, dim3(1), dim3(256), 0, s, tr->logits, tr->labels, N, (const AdamHp*)tr->hp, tr->loss, tr->dz);

    // ---------------- backward: classifier
    const float* dy = tr->dz;
    for (int k = 3; k >= 0; k--) {
        TrainFc& f = tr->fc[k];
        const float* xin_k = k == 0 ? pooled : tr->fc[k - 1].y;
        hipLaunchKernelGGL(lgt_fc_bwd_w_kernel, dim3(cdiv(f.in, 64), f.out), dim3(64, kRG), 0, s, dy, xin_k, tr->G + f.w,
                           tr->G + f.b, N, f.in, f.out);
        float* dxk = tr->dfc[0];   // never the buffer dy lives in (dz or dfc[1]): the kernel reads all of dy per output
        hipLaunchKernelGGL(lgt_fc_bwd_x_kernel, dim3(cdiv(f.in, 64), N), dim3(64, kRG), 0, s, dy, tr->P_ + f.w, dxk, N, f.in, f.out);
        if (k > 0) {
            TrainFc& p = tr->fc[k - 1];   // dxk = gradient at p.y -> through dropout / ReLU / BN1d to p.u
            float* du = tr->dfc[1];
            hipLaunchKernelGGL(lgt_bn1d_bwd_kernel, dim3(cdiv(p.out, 64)), dim3(64, kRG), 0, s, dxk, p.u, tr->mean + p.st,
                               tr->rstd + p.st, tr->P_ + p.g, tr->P_ + p.be, tr->masks + (size_t)N * p.mask, du, tr->G + p.g,
                               tr->G + p.be, N, p.out);
            dy = du;
        } else {
            dy = dxk;   // gradient at the pooled features [N][F]
        }
    }
    // attention + average pool
    const TrainLayer& last = tr->layers.back();
    const float* dm = nullptr;
    if (hid) {   // dy = gradient at g2 = ca * gs
        const float* m = spatial ? tr->ca_m : tr->gap;
        const size_t nh = (size_t)N * hid, nf = (size_t)N * F;
        hipLaunchKernelGGL(lgt_catt_gate_bwd_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, s, dy, tr->ca_ca, tr->gap, tr->ca_dgs,
                           tr->ca_dz2, nf);
        hipLaunchKernelGGL(lgt_fc_bwd_w_kernel, dim3(cdiv(hid, 64), F), dim3(64, kRG), 0, s, tr->ca_dz2, tr->ca_r,
                           tr->G + tr->ca_w2, tr->G + tr->ca_b2, N, hid, F);
        hipLaunchKernelGGL(lgt_fc_bwd_x_kernel, dim3(cdiv(hid, 64), N), dim3(64, kRG), 0, s, tr->ca_dz2, tr->P_ + tr->ca_w2,
                           tr->ca_dr, N, hid, F);
        hipLaunchKernelGGL(lgt_relu_bwd_kernel, dim3(cdiv(nh, 256)), dim3(256), 0, s, tr->ca_dr, tr->ca_z1, tr->ca_dz1, nh);
        hipLaunchKernelGGL(lgt_fc_bwd_w_kernel, dim3(cdiv(F, 64), hid), dim3(64, kRG), 0, s, tr->ca_dz1, m, tr->G + tr->ca_w1,
                           tr->G + tr->ca_b1, N, F, hid);
        hipLaunchKernelGGL(lgt_fc_bwd_x_kernel, dim3(cdiv(F, 64), N), dim3(64, kRG), 0, s, tr->ca_dz1, tr->P_ + tr->ca_w1,
                           tr->ca_dm, N, F, hid);
        dy = tr->ca_dgs;
        dm = tr->ca_dm;
    }
    hipLaunchKernelGGL(lgt_att_bwd_kernel, dim3(N), dim3(256), 0, s, last.out, tr->att_a, dy, tr->P_ + tr->att_w, spatial, F, P,
                       tr->dA[0], tr->att_part, dm);
    if (spatial)
        hipLaunchKernelGGL(lgt_colsum_kernel, dim3(cdiv(F + 1, 64)), dim3(64, kRG), 0, s, tr->att_part, N, F + 1, tr->G + tr->att_w);
    // encoder.  Main chain per layer: BN backward (reduce, dx) -> backward-data convolution; the backward-weights
    // convolution of the layer (+ the ordered sum of its pixel slices) runs beside it on the second stream.  dX is double
    // buffered: layer li's dX is rewritten by layer li-2, which first waits for layer li's backward-weights.
    hipStream_t sw = tr->side_stream ? tr->stream_w : s;
    int cur = 0;
    const int nl = (int)tr->layers.size();
    for (int li = nl - 1; li >= 0; li--) {
        TrainLayer& L = tr->layers[li];
        const float* ain = li == 0 ? tr->xin : tr->layers[li - 1].out;
        const float* dm = L.pool ? tr->masks + (size_t)N * L.mask : nullptr;
        const float *mu = tr->mean + L.st, *rs = tr->rstd + L.st, *g = tr->P_ + L.g, *be = tr->P_ + L.be;
        float* dXl = tr->dX[li & 1];
        if (tr->side_stream && li + 2 < nl) TR_HIP(hipStreamWaitEvent(s, tr->ev_wg[li + 2], 0));
        const int S = bn_chunks(N, L.co, L.wi * L.wi), chunk = (N + S - 1) / S, Sy = (N + chunk - 1) / chunk;
        if (L.pool) {
            hipLaunchKernelGGL(lgt_bn_bwd_reduce_kernel<true>, dim3(L.co, Sy), dim3(256), 0, s, L.x, tr->dA[cur], mu, rs, g, be, dm,
                               tr->G + L.g, tr->G + L.be, tr->bn_part, chunk, dXl, N, L.co, L.wi);
        } else {
            hipLaunchKernelGGL(lgt_bn_bwd_reduce_kernel<false>, dim3(L.co, Sy), dim3(256), 0, s, L.x, tr->dA[cur], mu, rs, g, be, dm,
                               tr->G + L.g, tr->G + L.be, tr->bn_part, chunk, dXl, N, L.co, L.wi);
        }
        if (Sy > 1) {   // large batches: ordered sum of the chunks, then dx over the whole chip
            hipLaunchKernelGGL(lgt_rowsum_kernel, dim3(cdiv(2 * L.co, 64)), dim3(64), 0, s, tr->bn_part, 2 * L.co, Sy, tr->G + L.g,
                               tr->G + L.be, L.co);
            if (L.pool) {
                const size_t n = (size_t)N * L.co * L.wi * L.wi / 4;
                hipLaunchKernelGGL(lgt_bn_bwd_dx_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, L.x, tr->dA[cur], mu, rs, g, be,
                                   dm, tr->G + L.g, tr->G + L.be, dXl, N, L.co, L.wi);
            } else {
                const size_t n = (size_t)N * L.co * L.wi * L.wi;
                hipLaunchKernelGGL(lgt_bn_bwd_dx_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, L.x, tr->dA[cur], mu, rs, g, be,
                                   dm, tr->G + L.g, tr->G + L.be, dXl, N, L.co, L.wi);
            }
        }
        if (tr->side_stream) {
            TR_HIP(hipEventRecord(tr->ev_dx[li], s));
            TR_HIP(hipStreamWaitEvent(sw, tr->ev_dx[li], 0));
        }
        const size_t nw = (size_t)9 * L.ci * L.co;
        const int ntiles = conv_tiles(L.wi, N, kWgTile), Sw = wgrad_split(ntiles, nw, (int)(cdiv(L.co, 64) * cdiv(L.ci, 32)));
        launch_wgrad(L.wi, dim3(Sw, cdiv(L.co, 64), cdiv(L.ci, 32)), sw, ain, dXl, tr->partial, N, L.ci, L.co, ntiles);
        hipLaunchKernelGGL(lgt_wreduce_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, sw, tr->partial, Sw, L.ci, L.co, tr->G + L.w);
        if (tr->side_stream) TR_HIP(hipEventRecord(tr->ev_wg[li], sw));
        if (li > 0) {   // backward-data: a convolution of dX with the mirrored, channel-swapped weights
            launch_conv(tr, L.wi, s, dXl, L.wpd, (const float*)nullptr, tr->dA[cur ^ 1], N, L.co, L.ci);
            cur ^= 1;
        }
    }
    if (tr->side_stream)
        for (int li = 0; li < std::min(2, nl); li++) TR_HIP(hipStreamWaitEvent(s, tr->ev_wg[li], 0));   // sw is in order
    // ---------------- optimizer
    hipLaunchKernelGGL(lgt_sumsq_kernel, dim3(kNormParts), dim3(256), 0, s, tr->G, tr->n_params, tr->norm_part);
    hipLaunchKernelGGL(lgt_norm_kernel, dim3(1), dim3(64), 0, s, tr->norm_part, kNormParts, apply_update ? 1 : 0, tr->state);
    if (apply_update)
        hipLaunchKernelGGL(lgt_adam_kernel, dim3(cdiv(tr->n_params, 256)), dim3(256), 0, s, tr->P_, tr->G, tr->M, tr->V,
                           tr->n_params, tr->hp, tr->state);
    TR_HIP(hipGetLastError());
    return LG_OK;
}

// One optimisation step on N samples.  x [N][9][32][32], labels [N] (0/1) and masks (NULL, or [N][mask_row] keep masks
// already scaled by 1/(1-p)) are DEVICE pointers; hp is a host struct.  apply_update = 0 computes loss and gradients only.
int lg_train_step(lg_trainer* tr, const float* x, const float* labels, int N, const float* masks, uint64_t seed,
                  const lg_train_hparams* hp, int apply_update, float* loss_host, float* grad_norm_host,
                  float* logits_dev) {
    if (!tr || !x || !labels || !hp) return LG_ERR_INVALID;
    if (N < 2 || N > tr->capN) { tr->err = "lg_train_step: N must be in [2, max_batch] (BatchNorm needs a batch)"; return LG_ERR_INVALID; }
    hipSetDevice(tr->device);
    hipStream_t s = tr->stream;
    TR_HIP(hipMemcpyAsync(tr->xin, x, (size_t)N * 9 * 1024 * 4, hipMemcpyDeviceToDevice, s));
    TR_HIP(hipMemcpyAsync(tr->labels, labels, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    const AdamHp h = {hp->lr, hp->beta1, hp->beta2, hp->eps, hp->weight_decay, hp->max_grad_norm, hp->pos_weight, 0.0f};
    TR_HIP(hipMemcpyAsync(tr->hp, &h, sizeof(h), hipMemcpyHostToDevice, s));
    TR_HIP(hipMemcpyAsync(tr->seed_dev, &seed, sizeof(seed), hipMemcpyHostToDevice, s));
    if (masks) TR_HIP(hipMemcpyAsync(tr->masks, masks, (size_t)N * tr->mask_row * 4, hipMemcpyDeviceToDevice, s));
    const uint64_t key = (uint64_t)N | ((uint64_t)(masks ? 0 : 1) << 32) | ((uint64_t)(apply_update ? 1 : 0) << 33);
    bool launched = false;
    if (tr->use_graph) {
        auto it = tr->graphs.find(key);
        if (it == tr->graphs.end()) {
            // ~75 short launches on two streams captured once per (N, mask source, update) and replayed
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = enqueue_step(tr, N, masks == nullptr, apply_update);
                const hipError_t ee = hipStreamEndCapture(s, &g);
                if (rc == LG_OK && ee == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess)
                    it = tr->graphs.emplace(key, ge).first;
                if (g) hipGraphDestroy(g);
            }
            if (it == tr->graphs.end()) {
                (void)hipGetLastError();
                tr->use_graph = false;   // capture not available: plain launches from here on
            }
        }
        if (it != tr->graphs.end()) {
            TR_HIP(hipGraphLaunch(it->second, s));
            launched = true;
        }
    }
    if (!launched) {
        const int rc = enqueue_step(tr, N, masks == nullptr, apply_update);
        if (rc != LG_OK) return rc;
    }
    if (apply_update) tr->steps++;
    if (logits_dev) TR_HIP(hipMemcpyAsync(logits_dev, tr->logits, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    if (loss_host || grad_norm_host) {
        float l = 0.f, st[4] = {0.f, 0.f, 0.f, 0.f};
        TR_HIP(hipMemcpyAsync(&l, tr->loss, 4, hipMemcpyDeviceToHost, s));
        TR_HIP(hipMemcpyAsync(st, tr->state, 16, hipMemcpyDeviceToHost, s));
        TR_HIP(hipStreamSynchronize(s));
        if (loss_host) *loss_host = l;
        if (grad_norm_host) *grad_norm_host = st[1];
    }
    return LG_OK;
}

// Data-parallel training: lg_train_step(apply_update = 0) leaves the rank's gradients in the flat vector below; the
// caller averages it over the ranks (RCCL all-reduce on the tensor wrapping this pointer) and calls lg_train_apply.
int lg_train_grad_buffer(lg_trainer* tr, float** dev_ptr, int64_t* n) {
    if (!tr || !dev_ptr) return LG_ERR_INVALID;
    *dev_ptr = tr->G;
    if (n) *n = (int64_t)tr->n_params;
    return LG_OK;
}

// clip_grad_norm_ + Adam on the gradient vector as it stands (train_model.py:256-258)
int lg_train_apply(lg_trainer* tr, const lg_train_hparams* hp, float* grad_norm_host) {
    if (!tr || !hp) return LG_ERR_INVALID;
    hipSetDevice(tr->device);
    hipStream_t s = tr->stream;
    const AdamHp h = {hp->lr, hp->beta1, hp->beta2, hp->eps, hp->weight_decay, hp->max_grad_norm, hp->pos_weight, 0.0f};
    TR_HIP(hipMemcpyAsync(tr->hp, &h, sizeof(h), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lgt_sumsq_kernel, dim3(kNormParts), dim3(256), 0, s, tr->G, tr->n_params, tr->norm_part);
    hipLaunchKernelGGL(lgt_norm_kernel, dim3(1), dim3(64), 0, s, tr->norm_part, kNormParts, 1, tr->state);
    hipLaunchKernelGGL(lgt_adam_kernel, dim3(cdiv(tr->n_params, 256)), dim3(256), 0, s, tr->P_, tr->G, tr->M, tr->V,
                       tr->n_params, tr->hp, tr->state);
    TR_HIP(hipGetLastError());
    tr->steps++;
    if (grad_norm_host) {
        float st[4] = {0.f, 0.f, 0.f, 0.f};
        TR_HIP(hipMemcpyAsync(st, tr->state, 16, hipMemcpyDeviceToHost, s));
        TR_HIP(hipStreamSynchronize(s));
        *grad_norm_host = st[1];
    }
    return LG_OK;
}

int lg_train_sync(lg_trainer* tr) {
    if (!tr) return LG_ERR_INVALID;
    hipSetDevice(tr->device);
    TR_HIP(hipStreamSynchronize(tr->stream));
    return LG_OK;
}

}  // extern "C"
