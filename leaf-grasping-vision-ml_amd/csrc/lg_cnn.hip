// GraspPointCNN forward on gfx950 (see lg_cnn.h).  Reference: scripts/utils/ml_grasp_optimizer/model.py:5-128.
//
// Conv layers: out[n][co][y][x] = relu(b[co] + sum_{ky,kx,ci} w[ky][kx][ci][co] * in[n][ci][y+ky-1][x+kx-1])
// as D = W * X on v_mfma_f32_32x32x2_f32 with M = 32 output channels, N = 32 consecutive pixels of the
// planar image, K = (tap, input channel) pairs.  A operand (weights) and B operand (im2col of an LDS
// staged input tile) are one VGPR each; lanes 0..31 carry k, lanes 32..63 carry k+1.
// Accumulator lane l holds pixel l&31 and channels (r&3)+8(r>>2)+4(l>>5): a store instruction writes 32
// consecutive pixels of one channel row = one 128-byte line.
#include "lg_cnn.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef LG_WS_EXP
#define LG_WS_EXP 0   // timing ablations of lg_wino_ws_kernel (wrong results!): 1 no A loads, 2 no B reads, 4 no transform, 8 no input
#endif

__device__ float lg_zero_pad[4];  // zero-initialised source for padded / out-of-range direct-to-LDS lanes

namespace {

struct LayerCfg { int cin, cinp, cout, wi; bool pool; };
constexpr LayerCfg kLayers[6] = {
    {9, 10, 64, 32, false}, {64, 64, 64, 32, true},   {64, 64, 128, 16, false},
    {128, 128, 128, 16, true}, {128, 128, 256, 8, false}, {256, 256, 256, 8, true}};

// WG = 256 threads = 4 waves; wave tile = 2 pixel-blocks x 2 channel-blocks (4 accumulators).
// PP = pixel-block pairs per workgroup, CP = channel-block pairs per workgroup, PP*CP == 4.
template <int CIN, int CINP, int COUT, int WI, bool POOL, int KC, int PP, int CP>
__global__ __launch_bounds__(256, 2) void lg_conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ out) {
    static_assert(PP * CP == 4, "4 waves per workgroup");
    constexpr int PBROWS = 32 / WI > 0 ? 32 / WI : 1;   // image rows per 32-pixel block (WI=32:1, 16:2, 8:4)
    constexpr int ROWS = 2 * PP * PBROWS;               // output rows per workgroup
    constexpr int TR = ROWS + 2;                        // staged input rows (halo 1)
    constexpr int TWID = WI + 2;
    constexpr int COUT_T = 64 * CP;                     // output channels per workgroup
    constexpr int IN_CH_STRIDE = TR * TWID;             // linear: the LDS image is filled by global_load_lds
    constexpr int IN_ELEMS = KC * TR * TWID;
    constexpr int NIN = (IN_ELEMS + 255) / 256;
    constexpr int IN_PAD = NIN * 256;                   // every lane of every load instruction has a slot
    constexpr int W4_ELEMS = 9 * KC * (COUT_T / 4);
    constexpr int NW4 = (W4_ELEMS + 255) / 256;
    constexpr int BUF = IN_PAD + NW4 * 256 * 4;         // floats per stage
    // ONE shared object (a second one makes hipcc drain vmcnt before every ds_read): [stage][input | weights]
    __shared__ __attribute__((aligned(16))) float s_buf[2 * BUF];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wp_i = wave % PP, wc_i = wave / PP;       // pixel-pair / channel-pair of this wave
    constexpr int BANDS = WI / ROWS > 0 ? WI / ROWS : 1;
    const int n = blockIdx.x / (BANDS * (COUT / COUT_T));
    const int rem = blockIdx.x % (BANDS * (COUT / COUT_T));
    const int band = rem % BANDS, ct = rem / BANDS;
    const int y0 = band * ROWS;                          // first output row of the workgroup
    const int co0 = ct * COUT_T;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // per-lane B-operand base offsets (within one channel plane of s_in) for the two pixel blocks
    const int p = lane & 31, kh = lane >> 5;
    int boff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int pb = wp_i * 2 + i;                      // pixel block inside the workgroup tile
        const int row = pb * PBROWS + p / WI, x = p % WI; // output pixel inside the tile
        boff[i] = row * TWID + x;                         // + ky*TWID + kx addresses the tap
    }
    const int aoff = (wc_i * 64) + (lane & 31);           // A-operand: channel inside the tile (+32 for block 1)

    const float* in_n = in + (size_t)n * CIN * WI * WI;
    // Direct-to-LDS staging (global_load_lds): chunk c+1 streams into the other LDS stage while the MFMA loop
    // works on chunk c; no staging registers, no LDS write pass, one barrier per chunk.  The LDS destination of
    // one wave instruction is wave-uniform base + lane * size, so both images are stored in load order; zero
    // padding of the conv (and the slots past the end) is fetched from a zeroed device word.
    // Source offsets are chunk-invariant apart from the channel base: computed once, so issuing a chunk costs a
    // few VALU instructions per load instead of ~25 (the two workgroups of a CU run in lock-step; address math
    // at every chunk boundary would leave the MFMA pipe idle on both).
    int off_in[NIN];   // offset inside the chunk's first channel plane group, or -1: zero padding / unused slot
    int ci_in[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int idx = t + 256 * j;
        const int ci = idx / (TR * TWID), r2 = idx % (TR * TWID);
        const int ry = r2 / TWID, rx = r2 % TWID;
        const int gy = y0 - 1 + ry, gx = rx - 1;
        const bool ok = idx < IN_ELEMS && gy >= 0 && gy < WI && gx >= 0 && gx < WI;
        off_in[j] = ok ? (ci * WI + gy) * WI + gx : -1;
        ci_in[j] = ci;
    }
    int off_w[NW4];    // offset into the packed weights for c0 = 0, or -1
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        const int idx = t + 256 * j;
        const int q = idx % (COUT_T / 4), rest = idx / (COUT_T / 4);
        const int ci = rest % KC, tap = rest / KC;
        off_w[j] = (idx < W4_ELEMS) ? (tap * CINP + ci) * COUT + co0 + 4 * q : -1;
    }
    auto issue_chunk = [&](int c0, int stage) {
        float* sb = s_buf + stage * BUF;
        const float* in_c = in_n + (size_t)c0 * WI * WI;
        const float* w_c = wp + (size_t)c0 * COUT;
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const bool ok = off_in[j] >= 0 && (CIN == CINP || c0 + ci_in[j] < CIN);
            const float* src = ok ? in_c + off_in[j] : lg_zero_pad;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + 256 * j + 64 * wave), 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NW4; j++) {
            const float* src = off_w[j] >= 0 ? w_c + off_w[j] : lg_zero_pad;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + IN_PAD + 4 * (256 * j + 64 * wave)),
                                             16, 0, 0);
        }
    };
    issue_chunk(0, 0);
    int stage = 0;
    for (int c0 = 0; c0 < CINP; c0 += KC, stage ^= 1) {
        __syncthreads();   // own loads landed (vmcnt(0)) + every wave is done with the other stage
        if (c0 + KC < CINP) issue_chunk(c0 + KC, stage ^ 1);
        const float* s_in = s_buf + stage * BUF;
        const float* s_w = s_in + IN_PAD;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int k0 = 0; k0 < KC; k0 += 2) {
                const int ci = k0 + kh;
                const float a0 = s_w[(tap * KC + ci) * COUT_T + aoff];
                const float a1 = s_w[(tap * KC + ci) * COUT_T + aoff + 32];
                const float b0 = s_in[ci * IN_CH_STRIDE + boff[0] + ky * TWID + kx];
                const float b1 = s_in[ci * IN_CH_STRIDE + boff[1] + ky * TWID + kx];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias + ReLU (+ 2x2 max pool), planar NCHW store
    constexpr int WO = POOL ? WI / 2 : WI;
    float* out_n = out + (size_t)n * COUT * WO * WO;
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + wc_i * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float bv = bias[co];
            if (!POOL) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int pb = wp_i * 2 + i;
                    const int pix = (y0 * WI) + pb * 32 + p;   // linear pixel index in the image plane
                    out_n[(size_t)co * WI * WI + pix] = fmaxf(acc[i][j][r] + bv, 0.0f);
                }
            } else {
                if (WI == 32) {
                    // the wave's two pixel blocks are image rows y0+2a, y0+2a+1
                    float v = fmaxf(acc[0][j][r], acc[1][j][r]);
                    v = fmaxf(v, __shfl_xor(v, 1, 64));
                    const int yo = (y0 + wp_i * 2) / 2, xo = p >> 1;
                    if ((p & 1) == 0) out_n[((size_t)co * WO + yo) * WO + xo] = fmaxf(v + bv, 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        float v = acc[i][j][r];
                        v = fmaxf(v, __shfl_xor(v, WI, 64));   // row pair inside the pixel block
                        v = fmaxf(v, __shfl_xor(v, 1, 64));
                        const int pb = wp_i * 2 + i;
                        const int row = y0 + pb * PBROWS + p / WI, x = p % WI;
                        if (((p / WI) & 1) == 0 && (x & 1) == 0)
                            out_n[((size_t)co * WO + (row >> 1)) * WO + (x >> 1)] = fmaxf(v + bv, 0.0f);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Layer 0 (9 input channels, 32x32, no pool) as a workgroup per (patch, 64 output channels) that walks the four 8-row
// bands of its patch: the 23 KB weight image is staged once instead of once per band and the next band's input streams
// into the other LDS stage while the current one is multiplied -- the generic kernel above, launched per band, spent
// 0.43 of this layer's time on the staging round trip of a 180-MFMA workgroup (MFMA busy 0.57).
template <int COUT>
__global__ __launch_bounds__(256, 2) void lg_conv0_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int CIN = 9, CINP = 10, WI = 32, ROWS = 8, TR = ROWS + 2, TWID = WI + 2, BANDS = WI / ROWS;
    constexpr int IN_CH_STRIDE = TR * TWID;
    constexpr int IN_ELEMS = CINP * TR * TWID;
    constexpr int NIN = (IN_ELEMS + 255) / 256;
    constexpr int IN_PAD = NIN * 256;
    constexpr int W4_ELEMS = 9 * CINP * (64 / 4);
    constexpr int NW4 = (W4_ELEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) float s_buf[2 * IN_PAD + NW4 * 256 * 4];   // [input stage 0 | stage 1 | weights]
    float* const s_w = s_buf + 2 * IN_PAD;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // pixel-block pair of this wave: band rows 2w, 2w+1
    const int n = blockIdx.x / (COUT / 64), co0 = (blockIdx.x % (COUT / 64)) * 64;
    const int p = lane & 31, kh = lane >> 5;
    const int boff0 = (2 * wave) * TWID + p, boff1 = (2 * wave + 1) * TWID + p;
    const int aoff = lane & 31;
    const float* in_n = in + (size_t)n * CIN * WI * WI;

    // staging slots: (channel, row of the band window, column); the row decides validity per band
    int off0[NIN], ry_[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int idx = t + 256 * j;
        const int ci = idx / (TR * TWID), r2 = idx % (TR * TWID);
        const int ry = r2 / TWID, gx = r2 % TWID - 1;
        const bool okx = idx < IN_ELEMS && ci < CIN && gx >= 0 && gx < WI;
        off0[j] = (ci * WI + ry - 1) * WI + gx;           // band 0; + ROWS * WI per band
        ry_[j] = okx ? ry : -1000;                        // never valid
    }
    auto issue_input = [&](int band, int stage) {
        float* sb = s_buf + stage * IN_PAD;
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const int gy = band * ROWS - 1 + ry_[j];
            const float* src = (gy >= 0 && gy < WI) ? in_n + off0[j] + band * ROWS * WI : lg_zero_pad;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + 256 * j + 64 * wave), 4, 0, 0);
        }
    };
#pragma unroll
    for (int j = 0; j < NW4; j++) {   // weights once: [tap][ci][64 channels], 16 bytes per lane
        const int idx = t + 256 * j;
        const int q = idx % 16, rest = idx / 16;
        const int ci = rest % CINP, tap = rest / CINP;
        const float* src = idx < W4_ELEMS ? wp + ((size_t)tap * CINP + ci) * COUT + co0 + 4 * q : lg_zero_pad;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(s_w + 4 * (256 * j + 64 * wave)), 16, 0, 0);
    }
    issue_input(0, 0);
    float bv[2][16];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) bv[j][r] = bias[co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];

#pragma unroll 1
    for (int band = 0; band < BANDS; band++) {
        __syncthreads();   // own loads landed (vmcnt(0)) + every wave is done with the other stage
        if (band + 1 < BANDS) issue_input(band + 1, (band + 1) & 1);
        const float* s_in = s_buf + (band & 1) * IN_PAD;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int k0 = 0; k0 < CINP; k0 += 2) {
                const int ci = k0 + kh;
                const float a0 = s_w[(tap * CINP + ci) * 64 + aoff];
                const float a1 = s_w[(tap * CINP + ci) * 64 + aoff + 32];
                const float b0 = s_in[ci * IN_CH_STRIDE + boff0 + ky * TWID + kx];
                const float b1 = s_in[ci * IN_CH_STRIDE + boff1 + ky * TWID + kx];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // bias + ReLU, one 128-byte row segment per (channel, row) and store instruction
        float* out_n = out + (size_t)n * COUT * WI * WI;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int y = band * ROWS + 2 * wave + i;
                    out_n[((size_t)co * WI + y) * WI + p] = fmaxf(acc[i][j][r] + bv[j][r], 0.0f);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) form of the same 3x3 convolutions (layers 1..5; layer 0 has 9 input channels and stays
// direct):  Y = A^T [ (G g G^T) .* (B^T d B) ] A  per 4x4 input tile d / 2x2 output tile Y, i.e. 16 independent
// contractions over the input channels instead of 36 multiply-adds per output pair: 2.25x fewer MFMA flops,
// still exact-f32 products and f32 accumulation (v_mfma_f32_16x16x4_f32).
//
// Workgroup = 256 threads = 32 tiles x 64 output channels x all 16 Winograd positions.  Wave w owns output
// channels 16w..16w+15: M = 16 channels, N = 16 tiles, K = 4 input channels per MFMA, 16 positions x 2 tile
// halves = 32 accumulators (128 VGPRs).  Every lane therefore ends with all 16 positions of its (channel,
// tile) pairs in its own registers and the output transform + bias + ReLU (+ the 2x2 max-pool, which is
// exactly one output tile) needs no exchange.
//   * A operand  U[ci][co][16]: 64 contiguous bytes per lane straight from L2 into registers (each element is
//     used by one wave only, LDS would add nothing); next k-step prefetched during the current one.
//   * B operand  V[ci][tile][16 (+4 pad)]: input chunk of 8 channels streamed by global_load_lds into a 2-stage
//     LDS ring, transformed by thread (ci, tile) = (t>>5, t&31) with 8 ds_read_b64 + 32 adds + 4 ds_write_b128,
//     read back as 4 ds_read_b128 per tile half (row stride 80 B: conflict free).
// WAVES = 4 (default): 256 threads, 64 output channels, chunks of 8 input channels, 2 workgroups per CU.
// WAVES = 8 (LG_CNN_WIDE=1, layers with >= 128 output channels): 512 threads, 128 output channels share one transformed
// tile block, chunks of 16 input channels, 1 workgroup per CU -- half the staging + transform work per MFMA.
template <int CIN, int COUT, int WI, bool POOL, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void lg_wino_kernel(const float* __restrict__ in, const float* __restrict__ U,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const float* __restrict__ zero_tail, int N, int ntb) {
    constexpr int THREADS = 64 * WAVES;
    constexpr int KC = 2 * WAVES;                          // input channels per chunk: one (channel, tile) item per thread
    constexpr int KS = KC / 4;                             // MFMA k-steps per chunk
    constexpr int CB = 16 * WAVES;                         // output channels per workgroup
    constexpr int TC = WI / 2, TP = TC * TC;               // tile columns, tiles per patch
    constexpr int PB = TP >= 32 ? 1 : 32 / TP;             // patches per workgroup (8x8 images: 2)
    constexpr int BPP = TP >= 32 ? TP / 32 : 1;            // workgroups (row bands) per patch
    constexpr int TPB = 32 / PB;                           // tiles of one patch inside the workgroup
    constexpr int TROWS = TPB / TC;                        // tile rows per band
    constexpr int RH = 2 * TROWS + 2, RW = WI + 2;         // staged input region (halo 1) per channel and patch
    constexpr int RS = RH * RW, S = PB * RS;
    constexpr int NIN = (KC * S + THREADS - 1) / THREADS;
    constexpr int VS = 20;                                 // floats per (ci, tile) row of V: 16 positions + 4 pad
    constexpr int NCB = COUT / CB;
    constexpr int NC = CIN / KC;
    static_assert(CIN % KC == 0 && COUT % CB == 0 && TPB % TC == 0 && (RW % 2) == 0 && (RS % 2) == 0 && KS % 2 == 0, "shape");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float s_mem[2 * NIN * THREADS + KC * 32 * VS];
    float* const s_v = s_mem + 2 * NIN * THREADS;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: LDS-DMA bases (M0) stay on the SALU
    // XCD-aware order: an XCD walks consecutive (tile block, channel block) pairs, so the NCB workgroups that
    // share an input band run back to back on one L2.
    long long id = blockIdx.x;
    {
        const long long total = (long long)ntb * NCB, q = total / 8, r = total % 8;
        const long long xcd = id % 8, j = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tb = (int)(id / NCB), cb = (int)(id % NCB);
    const int n0 = PB > 1 ? tb * PB : tb / BPP;
    const int band = PB > 1 ? 0 : tb % BPP;
    const int y0 = band * 2 * TROWS;                       // first output row of the band
    const int co0 = cb * CB;

    // ---- input staging: chunk-invariant 32-bit byte offsets from a wave-uniform base that advances per chunk
    //      (global_load_lds saddr + voffset form: issuing a chunk costs ~3 instructions per load).  Conv padding and
    //      unused slots read zeros from `zero_tail`, a zeroed region >= CIN*WI*WI floats behind the activation buffer:
    //      their offsets advance with the base like everybody else's and stay inside it.
    const float* in_n = in + (size_t)n0 * CIN * WI * WI;
    const unsigned tail_rel = (unsigned)((const char*)zero_tail - (const char*)in_n);
    unsigned voff[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int e = t + THREADS * j;
        const int ci = e / S, r = e % S;
        const int pb = r / RS, r2 = r % RS;
        const int ry = r2 / RW, rx = r2 % RW;
        const int gy = y0 - 1 + ry, gx = rx - 1;
        const bool ok = e < KC * S && gy >= 0 && gy < WI && gx >= 0 && gx < WI && n0 + pb < N;
        voff[j] = ok ? 4u * (unsigned)(((pb * CIN + ci) * WI + gy) * WI + gx) : tail_rel;
    }
    auto issue_input = [&](int c, int stage) {
        const char* in_c = (const char*)(in_n + (size_t)c * KC * WI * WI);
        float* sb = s_mem + stage * (NIN * THREADS);
#pragma unroll
        for (int j = 0; j < NIN; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in_c + voff[j]),
                                             (__attribute__((address_space(3))) void*)(sb + THREADS * j + 64 * wave), 4, 0, 0);
    };
    // ---- A operand: lane (co = lane & 15, k = lane >> 4) reads the 16 positions of U[ci][co]
    const float* u_lane = U + ((size_t)(lane >> 4) * COUT + co0 + 16 * wave + (lane & 15)) * 16;
    auto load_u = [&](int ks, f32x4 (&a)[4]) {             // ks = global k-step (4 input channels each)
        const f32x4* p = reinterpret_cast<const f32x4*>(u_lane + (size_t)ks * 4 * COUT * 16);
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = p[q];
    };
    // ---- transform role: (channel of the chunk, tile)
    const int tci = t >> 5, tau = t & 31;
    const int tpb = tau / TPB, ttl = tau % TPB;
    const int ttr = ttl / TC, ttc = ttl % TC;
    const int tsrc = tci * S + tpb * RS + (2 * ttr) * RW + 2 * ttc;
    float* const tdst = s_v + (tci * 32 + tau) * VS;
    const float* const vsrc = s_v + ((lane >> 4) * 32 + (lane & 15)) * VS;

    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; p++)
#pragma unroll
        for (int h = 0; h < 2; h++) acc[p][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // one k-step = 32 MFMAs; `mid` runs after the first 8 (i.e. after the wait for this step's A operand, so a
    // prefetch issued there is not caught by that wait)
    auto mfma_step = [&](int ksl, const f32x4 (&a)[4], auto&& mid) {   // ksl = k-step inside the chunk
        const float* vb = vsrc + ksl * 4 * 32 * VS;
        f32x4 b0[4], b1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            b0[q] = *reinterpret_cast<const f32x4*>(vb + 4 * q);
            b1[q] = *reinterpret_cast<const f32x4*>(vb + 16 * VS + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[4 * q + e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][e], b0[q][e], acc[4 * q + e][0], 0, 0, 0);
                acc[4 * q + e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][e], b1[q][e], acc[4 * q + e][1], 0, 0, 0);
            }
            if (q == 0) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    f32x4 ab[2][4];   // A operand ring: k-step ksl of a chunk uses ab[ksl & 1]
    issue_input(0, 0);
    load_u(0, ab[0]);
#pragma unroll 1
    for (int c = 0; c < NC; c++) {
        __syncthreads();                                   // vmcnt(0): input(c) + ab[0] landed; V / other stage free
        load_u(KS * c + 1, ab[1]);
        if (c + 1 < NC) issue_input(c + 1, (c + 1) & 1);
        {   // V = B^T d B for (tci, tau)
            const float* sp = s_mem + (c & 1) * (NIN * THREADS) + tsrc;
            // packed-f32 form: rows as (lo, hi) pairs; the column step folds its negations / half selections into
            // v_pk_add_f32 modifiers instead of moves
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // plain float reads (merged into ds_read2_b64): a float2-typed read makes hipcc drain vmcnt, i.e. the
                // next chunk's global_load_lds prefetch, in front of it
                dl[i] = (f32x2){sp[i * RW], sp[i * RW + 1]};
                dh[i] = (f32x2){sp[i * RW + 2], sp[i * RW + 3]};
            }
            f32x2 tl[4], th[4];
            tl[0] = dl[0] - dl[2]; th[0] = dh[0] - dh[2];
            tl[1] = dl[1] + dl[2]; th[1] = dh[1] + dh[2];
            tl[2] = dl[2] - dl[1]; th[2] = dh[2] - dh[1];
            tl[3] = dl[1] - dl[3]; th[3] = dh[1] - dh[3];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // column step in two instructions per row (hipcc does not fold the half selections into op_sel itself):
                //   v01 = (t0 - t2, t1 + t2): src1 = th.lo for both lanes, negated in the low lane
                //   v23 = (t2 - t1, t1 - t3): src0 = th, src1 = tl.hi for both lanes; low lane -src1, high lane -src0
                f32x2 v01, v23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"
                    : "=v"(v01) : "v"(tl[i]), "v"(th[i]));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]"
                    : "=v"(v23) : "v"(th[i]), "v"(tl[i]));
                *reinterpret_cast<f32x2*>(tdst + 4 * i) = v01;
                *reinterpret_cast<f32x2*>(tdst + 4 * i + 2) = v23;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // V visible; does not drain the prefetches
        // A operand of k-step ksl + 1 (the next chunk's first one at the end), requested behind the first MFMAs of step
        // ksl >= 1 into the buffer step ksl - 1 has released: hipcc waits with vmcnt(0) for the buffer in use, and sinks
        // an unpinned prefetch to the loop end where the barrier's vmcnt(0) exposes it
#pragma unroll
        for (int ksl = 0; ksl < KS; ksl++)
            mfma_step(ksl, ab[ksl & 1], [&] {
                if (ksl >= 1) {
                    const int ksn = KS * c + ksl + 1;
                    load_u(ksn < KS * NC ? ksn : KS * NC - 1, ab[(ksl + 1) & 1]);
                }
            });
    }

    // ---- epilogue: Y = A^T M A, bias, ReLU (+ max over the 2x2 tile), planar NCHW store
    constexpr int WO = POOL ? WI / 2 : WI;
    const int etau0 = lane & 15;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int etau = 16 * h + etau0;
        const int epb = etau / TPB, etl = etau % TPB;
        const int n = n0 + epb;
        const int trg = band * TROWS + etl / TC, tcg = etl % TC;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int co = co0 + 16 * wave + 4 * (lane >> 4) + r;
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float m0 = acc[j][h][r], m1 = acc[4 + j][h][r], m2 = acc[8 + j][h][r], m3 = acc[12 + j][h][r];
                s0[j] = m0 + m1 + m2;
                s1[j] = m1 - m2 - m3;
            }
            const float bv = bias[co];
            const float y00 = s0[0] + s0[1] + s0[2] + bv, y01 = s0[1] - s0[2] - s0[3] + bv;
            const float y10 = s1[0] + s1[1] + s1[2] + bv, y11 = s1[1] - s1[2] - s1[3] + bv;
            if (n < N) {
                float* o = out + (size_t)n * COUT * WO * WO + (size_t)co * WO * WO;
                if (POOL) {
                    o[trg * WO + tcg] = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)), 0.0f);
                } else {
                    *reinterpret_cast<float2*>(o + (2 * trg) * WO + 2 * tcg) = make_float2(fmaxf(y00, 0.f), fmaxf(y01, 0.f));
                    *reinterpret_cast<float2*>(o + (2 * trg + 1) * WO + 2 * tcg) = make_float2(fmaxf(y10, 0.f), fmaxf(y11, 0.f));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Producer / consumer form of the Winograd layer (LG_CNN_WS_KC=8; measured SLOWER than the lock-step kernel above, kept
// as the documented experiment with its ablation switches).  The kernel above alternates "transform a
// chunk" and "64 MFMAs" in every wave, so the matrix pipe idles whenever both workgroups of a CU transform or
// wait at a barrier together (measured 0.57-0.60 of the f32 MFMA peak).  Here a persistent workgroup of 512
// threads per CU is split by role:
//   waves 4..7 (producers): stream the input chunks by global_load_lds and write V = B^T d B into a 2-stage LDS ring,
//                           one chunk ahead of the consumers, across work-item boundaries;
//   waves 0..3 (consumers): nothing but operand fetches and MFMAs (16 output channels x 32 tiles x 16 positions
//                           each, as above), B operands of the next half k-step requested before the MFMAs of the
//                           current one, A operands of the next k-step prefetched from L2;
// one s_barrier per chunk hands a V stage over (consumers arrive with an LDS-only wait, so their A prefetch stays
// in flight).  A work item = (32-tile block, 64-channel block); a workgroup takes a contiguous range of items, so the
// channel blocks of one input band follow each other on one CU (input band hot in L2) and prologue / epilogue
// latencies are paid once per workgroup / overlapped by the producers' run-ahead.
template <int CIN, int COUT, int WI, bool POOL, int KC>
__global__ __launch_bounds__(512, 1) void lg_wino_ws_kernel(const float* __restrict__ in, const float* __restrict__ U,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int N, int ntb) {
    constexpr int TC = WI / 2, TP = TC * TC;
    constexpr int PB = TP >= 32 ? 1 : 32 / TP;
    constexpr int BPP = TP >= 32 ? TP / 32 : 1;
    constexpr int TPB = 32 / PB;
    constexpr int TROWS = TPB / TC;
    constexpr int RH = 2 * TROWS + 2, RW = WI + 2;
    constexpr int RS = RH * RW, S = PB * RS;
    constexpr int NIN = (KC * S + 255) / 256;
    constexpr int INB = NIN * 256;                         // floats per input stage
    constexpr int VS = 20;
    constexpr int VB = KC * 32 * VS;                       // floats per V stage
    constexpr int NCB = COUT / 64;
    constexpr int NC = CIN / KC;                           // chunks per work item
    constexpr int KS = KC / 4;                             // MFMA k-steps per chunk
    static_assert(CIN % KC == 0 && COUT % 64 == 0 && TPB % TC == 0 && (KS % 2) == 0 && KC % 8 == 0, "shape");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int RI = 3;                                  // input ring: chunks sq+2, sq+3 in flight while sq+1 is transformed
    __shared__ __attribute__((aligned(16))) float s_mem[RI * INB + 2 * VB];
    float* const s_v = s_mem + RI * INB;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const long long total = (long long)ntb * NCB;
    const int it0 = (int)(total * blockIdx.x / gridDim.x), it1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
    const int nitems = it1 - it0;
    if (nitems <= 0) return;
    const int nseq = nitems * NC;                          // chunks this workgroup streams

    if (wave >= 4) {
        // ================================================================= producers
        const int pt = t - 256, pw = wave - 4;
        int off_in[NIN];
        const float* in_n = in;
        auto issue = [&](int sq) {
            const int c = sq % NC;
            if (c == 0) {                                  // new work item: staging offsets of its band
                const int tb = (it0 + sq / NC) / NCB;
                const int n0 = PB > 1 ? tb * PB : tb / BPP;
                const int y0 = (PB > 1 ? 0 : tb % BPP) * 2 * TROWS;
#pragma unroll
                for (int j = 0; j < NIN; j++) {
                    const int e = pt + 256 * j;
                    const int ci = e / S, r = e % S;
                    const int pb = r / RS, r2 = r % RS;
                    const int ry = r2 / RW, rx = r2 % RW;
                    const int gy = y0 - 1 + ry, gx = rx - 1;
                    const bool ok = e < KC * S && gy >= 0 && gy < WI && gx >= 0 && gx < WI && n0 + pb < N;
                    off_in[j] = ok ? ((pb * CIN + ci) * WI + gy) * WI + gx : -1;
                }
                in_n = in + (size_t)n0 * CIN * WI * WI;
            }
            const float* in_c = in_n + (size_t)c * KC * WI * WI;
            float* sb = s_mem + (sq % RI) * INB;
#pragma unroll
            for (int j = 0; j < NIN; j++) {
                const float* src = off_in[j] >= 0 ? in_c + off_in[j] : lg_zero_pad;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(sb + 256 * j + 64 * pw), 4, 0, 0);
            }
        };
        const int tau = pt & 31;
        const int tpb = tau / TPB, ttl = tau % TPB;
        const int tsrc0 = tpb * RS + (2 * (ttl / TC)) * RW + 2 * (ttl % TC);
        auto transform = [&](int sq) {                     // V = B^T d B for (channel, tile) = (pt >> 5 [+8], pt & 31)
#pragma unroll
            for (int it = 0; it < KC / 8; it++) {
                const int tci = (pt >> 5) + 8 * it;
                const float* sp = s_mem + (sq % RI) * INB + tci * S + tsrc0;
                float d[4][4];
#pragma unroll
                for (int i = 0; i < 4; i++) {              // plain float reads (see lg_wino_kernel)
                    d[i][0] = sp[i * RW]; d[i][1] = sp[i * RW + 1]; d[i][2] = sp[i * RW + 2]; d[i][3] = sp[i * RW + 3];
                }
                float r[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    r[0][j] = d[0][j] - d[2][j];
                    r[1][j] = d[1][j] + d[2][j];
                    r[2][j] = d[2][j] - d[1][j];
                    r[3][j] = d[1][j] - d[3][j];
                }
                float* dst = s_v + (sq & 1) * VB + (tci * 32 + tau) * VS;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f32x4 v = {r[i][0] - r[i][2], r[i][1] + r[i][2], r[i][2] - r[i][1], r[i][1] - r[i][3]};
                    *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
                }
            }
        };
        // HBM latency (2-3 us) exceeds one chunk of MFMA time (~1 us): the input runs two chunks ahead of the transform.
        // Loads return in order, so "at most NIN outstanding" = everything but the newest chunk has landed.
        issue(0);
        if (1 < nseq) issue(1);
        if (2 < nseq) {
            issue(2);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NIN) : "memory");   // input(0) visible
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        transform(0);
        if (2 < nseq) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NIN) : "memory");  // V(0), input(1)
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
        for (int sq = 0; sq < nseq; sq++) {                // consumers work on V(sq)
            const bool more_in = sq + 3 < nseq;
            if (more_in && !(LG_WS_EXP & 8)) issue(sq + 3);   // stage sq % 3: its last reader was transform(sq)
            if (sq + 1 < nseq && !(LG_WS_EXP & 4)) transform(sq + 1);          // V stage (sq+1) & 1: its last readers were the MFMAs of sq-1
            if (more_in) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NIN) : "memory");  // input(sq+2) landed
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        // ================================================================= consumers
        const float* const vsrc = s_v + ((lane >> 4) * 32 + (lane & 15)) * VS;
        const float* const u_lane = U + ((size_t)(lane >> 4) * COUT + 16 * wave + (lane & 15)) * 16;
        f32x4 acc[16][2];
        f32x4 ab[2][4];                                    // A operand ring (k-step parity)
        f32x4 bb[2][4];                                    // B operand ring (half k-step parity)
        auto load_u = [&](int cb, int ks, f32x4 (&a)[4]) { // ks = k-step inside the item (4 input channels each)
            const f32x4* p = reinterpret_cast<const f32x4*>(u_lane + ((size_t)ks * 4 * COUT + cb * 64) * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) a[q] = p[q];
        };
        auto load_b = [&](int stage, int ksl, int h, f32x4 (&b)[4]) {
            const float* vb = vsrc + stage * VB + (ksl * 4 * 32 + 16 * h) * VS;
#pragma unroll
            for (int q = 0; q < 4; q++) b[q] = *reinterpret_cast<const f32x4*>(vb + 4 * q);
        };
        // one chunk: KS k-steps x 2 tile halves; `first` zero-initialises the accumulators with the first k-step
        auto chunk = [&](auto first_tag, int stage, int cb, int c, int cb_next, bool more) {
            constexpr bool FIRST = decltype(first_tag)::value;
            load_b(stage, 0, 0, bb[0]);
            if (LG_WS_EXP & 2) load_b(stage, 0, 1, bb[1]);
#pragma unroll
            for (int ksl = 0; ksl < KS; ksl++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int u = 2 * ksl + h;
                    if (u + 1 < 2 * KS && !(LG_WS_EXP & 2)) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_b(stage, (u + 1) >> 1, (u + 1) & 1, bb[(u + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (FIRST && ksl == 0)
                                acc[4 * q + e][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    ab[ksl & 1][q][e], bb[u & 1][q][e], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else
                                acc[4 * q + e][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[ksl & 1][q][e], bb[u & 1][q][e],
                                                                                           acc[4 * q + e][h], 0, 0, 0);
                        }
                        if (q == 0 && h == 0) {
                            // A operand of the next k-step (next chunk / next item at the end), behind the wait for
                            // this step's A so that wait cannot catch it
                            __builtin_amdgcn_sched_barrier(0);
                            // (unconditional, clamped at the very end: a branch here makes hipcc wait vmcnt(0) at the join)
                            const int ksn = c * KS + ksl + 1;
                            const bool within = ksn < NC * KS;
                            if (!(LG_WS_EXP & 1)) load_u(within || !more ? cb : cb_next, within ? ksn : 0, ab[(ksl + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // done with this V stage
        };

        load_u(it0 % NCB, 0, ab[0]);
        if (LG_WS_EXP & 1) load_u(it0 % NCB, 1, ab[1]);
        asm volatile("s_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        constexpr int WO = POOL ? WI / 2 : WI;
        int sq = 0;
#pragma unroll 1
        for (int item = it0; item < it1; item++) {
            const int tb = item / NCB, cb = item % NCB;
            const int cb_next = (item + 1) % NCB;
            const bool more = item + 1 < it1;
            const int co_l = cb * 64 + 16 * wave + 4 * (lane >> 4);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co_l);
            chunk(std::true_type{}, sq & 1, cb, 0, cb_next, more);
            sq++;
#pragma unroll 1
            for (int c = 1; c < NC; c++, sq++) chunk(std::false_type{}, sq & 1, cb, c, cb_next, more);

            // ---- epilogue: Y = A^T M A, bias, ReLU (+ max over the 2x2 tile), planar NCHW store
            const int n0 = PB > 1 ? tb * PB : tb / BPP;
            const int band = PB > 1 ? 0 : tb % BPP;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int etau = 16 * h + (lane & 15);
                const int epb = etau / TPB, etl = etau % TPB;
                const int n = n0 + epb;
                const int trg = band * TROWS + etl / TC, tcg = etl % TC;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float s0[4], s1[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float m0 = acc[j][h][r], m1 = acc[4 + j][h][r], m2 = acc[8 + j][h][r], m3 = acc[12 + j][h][r];
                        s0[j] = m0 + m1 + m2;
                        s1[j] = m1 - m2 - m3;
                    }
                    const float y00 = s0[0] + s0[1] + s0[2] + bv[r], y01 = s0[1] - s0[2] - s0[3] + bv[r];
                    const float y10 = s1[0] + s1[1] + s1[2] + bv[r], y11 = s1[1] - s1[2] - s1[3] + bv[r];
                    if (n < N) {
                        float* o = out + (size_t)n * COUT * WO * WO + (size_t)(co_l + r) * WO * WO;
                        if (POOL) {
                            o[trg * WO + tcg] = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)), 0.0f);
                        } else {
                            *reinterpret_cast<float2*>(o + (2 * trg) * WO + 2 * tcg) = make_float2(fmaxf(y00, 0.f), fmaxf(y01, 0.f));
                            *reinterpret_cast<float2*>(o + (2 * trg + 1) * WO + 2 * tcg) = make_float2(fmaxf(y10, 0.f), fmaxf(y11, 0.f));
                        }
                    }
                }
            }
        }
    }
}

// attention, global average pool, classifier F -> F -> F/2 -> F/4 -> 1 (BN folded).  model.py:30-60,63-84,108-128.
//   spatial: x * sigmoid(conv1x1(x) F->1)            channel: x * sigmoid(W2 relu(W1 gap(x) + b1) + b2), hidden F/16
//   hybrid : x * spatial(x) * channel(x)              none   : x
// One workgroup per patch; thread t holds channels t and t + 256 (F <= 512) with their npix (16 or 4) pixels.
// h is [n][Cp][npix] (Cp = channel count padded to the conv kernels' 64-channel granule).
template <int F, int npix>   // compile-time sizes: the channel / pixel loops unroll (generic runtime sizes cost 40 % more)
__global__ __launch_bounds__(256) void lg_head_kernel(const float* __restrict__ h, int Cp, int att_type,
                                                      const float* __restrict__ att_w, float att_b,
                                                      const float* __restrict__ ca_w1, const float* __restrict__ ca_b1,
                                                      const float* __restrict__ ca_w2, const float* __restrict__ ca_b2,
                                                      const float* __restrict__ w0,
                                                      const float* __restrict__ b0, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, const float* __restrict__ w3,
                                                      const float* __restrict__ b3, float* __restrict__ logits) {
    __shared__ float s_a[4][32];
    __shared__ float s_f[512], s_g[512];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* hn = h + (size_t)n * Cp * npix;
    float v[2][16];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int c = t + 256 * i;
#pragma unroll
        for (int p = 0; p < 16; p++) v[i][p] = (c < F && p < npix) ? hn[(size_t)c * npix + p] : 0.0f;
    }
    const bool spatial = att_type == LG_ATT_SPATIAL || att_type == LG_ATT_HYBRID;
    const bool channel = att_type == LG_ATT_CHANNEL || att_type == LG_ATT_HYBRID;
    const float inv_np = 1.0f / (float)npix;
    // sum over all channels of a per-thread value, for up to 32 slots, through s_a (waves) -- result read by everyone
    float f[2] = {0.0f, 0.0f};   // mean over the pixels of x (* spatial attention)
    if (spatial) {
        const float aw0 = t < F ? att_w[t] : 0.0f, aw1 = t + 256 < F ? att_w[t + 256] : 0.0f;
        for (int p = 0; p < npix; p++) {
            float s = aw0 * v[0][p] + aw1 * v[1][p];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) s_a[wave][p] = s;
        }
        __syncthreads();
        for (int p = 0; p < npix; p++) {
            const float z = s_a[0][p] + s_a[1][p] + s_a[2][p] + s_a[3][p] + att_b;
            const float a = 1.0f / (1.0f + expf(-z));
            f[0] += v[0][p] * a;
            f[1] += v[1][p] * a;
        }
    } else {
        for (int p = 0; p < npix; p++) { f[0] += v[0][p]; f[1] += v[1][p]; }
    }
    f[0] *= inv_np; f[1] *= inv_np;
    if (channel) {   // squeeze-and-excitation on the un-attended x: gap -> F/16 -> F -> sigmoid
        const int hid = F / 16;
        float g[2] = {0.0f, 0.0f};
        for (int p = 0; p < npix; p++) { g[0] += v[0][p]; g[1] += v[1][p]; }
        g[0] *= inv_np; g[1] *= inv_np;
        __syncthreads();   // s_a is reused
        for (int j = 0; j < hid; j++) {
            float s = (t < F ? ca_w1[j * F + t] * g[0] : 0.0f) + (t + 256 < F ? ca_w1[j * F + t + 256] * g[1] : 0.0f);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) s_a[wave][j] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int c = t + 256 * i;
            if (c < F) {
                float e = ca_b2[c];
                for (int j = 0; j < hid; j++) {
                    const float z = fmaxf(s_a[0][j] + s_a[1][j] + s_a[2][j] + s_a[3][j] + ca_b1[j], 0.0f);
                    e += ca_w2[c * hid + j] * z;
                }
                f[i] *= 1.0f / (1.0f + expf(-e));
            }
        }
    }
    s_f[t] = f[0]; s_f[t + 256] = f[1];
    __syncthreads();
    // classifier: weights transposed [in][out], BN folded
    const int d1 = F, d2 = F / 2, d3 = F / 4;
    for (int o = t; o < d1; o += 256) {
        float s = b0[o];
        for (int c = 0; c < F; c++) s += w0[c * d1 + o] * s_f[c];
        s_g[o] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    for (int o = t; o < d2; o += 256) {
        float s = b1[o];
        for (int c = 0; c < d1; c++) s += w1[c * d2 + o] * s_g[c];
        s_f[o] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    for (int o = t; o < d3; o += 256) {
        float s = b2[o];
        for (int c = 0; c < d2; c++) s += w2[c * d3 + o] * s_f[c];
        s_g[o] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    {
        float s = t < d3 ? w3[t] * s_g[t] : 0.0f;   // d3 <= 128
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) s_a[wave][0] = s;
        __syncthreads();
        if (t == 0) logits[n] = s_a[0][0] + s_a[1][0] + s_a[2][0] + s_a[3][0] + b3[0];
    }
}

template <int L, int KC, int PP, int CP>
void launch_conv(const float* in, const LgCnn* c, float* out, int N, hipStream_t s) {
    constexpr LayerCfg cfg = kLayers[L];
    constexpr int PBROWS = 32 / cfg.wi > 0 ? 32 / cfg.wi : 1;
    constexpr int ROWS = 2 * PP * PBROWS;
    constexpr int BANDS = cfg.wi / ROWS > 0 ? cfg.wi / ROWS : 1;
    static_assert(ROWS <= cfg.wi && cfg.wi % ROWS == 0, "tile rows must divide the image");
    const int grid = N * BANDS * (cfg.cout / (64 * CP));
    hipLaunchKernelGGL((lg_conv3x3_kernel<cfg.cin, cfg.cinp, cfg.cout, cfg.wi, cfg.pool, KC, PP, CP>), dim3(grid), dim3(256), 0,
                       s, in, c->wconv[L], c->bconv[L], out);
}

// layer 0 of any encoder (9 input channels -> 64 or 128 padded output channels at 32x32): direct implicit GEMM,
// one workgroup per (patch, 64 channels) walking the four bands (LG_CNN_L0_BANDS=1: the per-band generic kernel, A/B)
template <int COUT>
void launch_conv0(const float* in, const LgCnn* c, float* out, int N, hipStream_t s) {
    if (getenv("LG_CNN_L0_BANDS")) {
        const int grid = N * 4 * (COUT / 64);   // PP = 4: 8 rows per workgroup -> 4 bands
        hipLaunchKernelGGL((lg_conv3x3_kernel<9, 10, COUT, 32, false, 10, 4, 1>), dim3(grid), dim3(256), 0, s, in, c->wconv[0],
                           c->bconv[0], out);
        return;
    }
    hipLaunchKernelGGL((lg_conv0_kernel<COUT>), dim3(N * (COUT / 64)), dim3(256), 0, s, in, c->wconv[0], c->bconv[0], out);
}

template <int L, int KC>
void launch_wino_ws(const float* in, const LgCnn* c, float* out, int N, hipStream_t s, int num_cu) {
    constexpr LayerCfg cfg = kLayers[L];
    constexpr int TP = (cfg.wi / 2) * (cfg.wi / 2);
    const int ntb = TP >= 32 ? N * (TP / 32) : (N + 32 / TP - 1) / (32 / TP);
    const long long total = (long long)ntb * (cfg.cout / 64);
    const int grid = (int)(total < num_cu ? total : num_cu);   // persistent: one workgroup per CU
    hipLaunchKernelGGL((lg_wino_ws_kernel<cfg.cin, cfg.cout, cfg.wi, cfg.pool, KC>), dim3(grid), dim3(512), 0, s, in,
                       c->uwino[L], c->bconv[L], out, N, ntb);
}

// Winograd layer shapes of the four encoder configurations of the reference's sweep (train_model_mlflow.py:177-182;
// 32-channel stages of 'lightweight' are zero-padded to the 64-channel granule): (cin, cout, width, pool)
#define LG_WINO_SHAPES(X)                                                                                          \
    X(64, 64, 32, true) X(64, 128, 16, false) X(128, 128, 16, true) X(128, 256, 8, false) X(256, 256, 8, true)     \
    X(64, 64, 16, false) X(64, 64, 16, true) X(64, 128, 8, false) X(128, 128, 8, true)                             \
    X(128, 128, 32, true) X(128, 256, 16, false) X(256, 256, 16, true) X(256, 512, 8, false) X(512, 512, 8, true)  \
    X(256, 512, 4, false) X(512, 512, 4, true)

bool wino_supported(int cin, int cout, int wi, bool pool) {
#define X(CI, CO, W_, P) if (cin == CI && cout == CO && wi == W_ && pool == P) return true;
    LG_WINO_SHAPES(X)
#undef X
    return false;
}

// 128 output channels per workgroup (WAVES = 8): the layers of the default encoder with >= 128 output channels
#define LG_WINO_WIDE_SHAPES(X) X(64, 128, 16, false) X(128, 128, 16, true) X(128, 256, 8, false) X(256, 256, 8, true)

bool launch_wino_rt(int cin, int cout, int wi, bool pool, const float* in, const float* U, const float* bias, float* out,
                    const float* zero_tail, int N, hipStream_t s) {
    const int tp = (wi / 2) * (wi / 2);
    const int ntb = tp >= 32 ? N * (tp / 32) : (N + 32 / tp - 1) / (32 / tp);
    if (getenv("LG_CNN_WIDE")) {
        const int gridw = ntb * (cout / 128);
#define X(CI, CO, W_, P)                                                                                           \
    if (cin == CI && cout == CO && wi == W_ && pool == P) {                                                        \
        hipLaunchKernelGGL((lg_wino_kernel<CI, CO, W_, P, 8>), dim3(gridw), dim3(512), 0, s, in, U, bias, out, zero_tail, \
                           N, ntb);                                                                                \
        return true;                                                                                               \
    }
        LG_WINO_WIDE_SHAPES(X)
#undef X
    }
    const int grid = ntb * (cout / 64);
#define X(CI, CO, W_, P)                                                                                           \
    if (cin == CI && cout == CO && wi == W_ && pool == P) {                                                        \
        hipLaunchKernelGGL((lg_wino_kernel<CI, CO, W_, P>), dim3(grid), dim3(256), 0, s, in, U, bias, out, zero_tail, N, \
                           ntb);                                                                                   \
        return true;                                                                                               \
    }
    LG_WINO_SHAPES(X)
#undef X
    return false;
}

}  // namespace

void lg_cnn_free(LgCnn* c) {
    auto F = [](float*& p) { if (p) hipFree(p); p = nullptr; };
    for (int i = 0; i < 8; i++) { F(c->wconv[i]); F(c->bconv[i]); F(c->uwino[i]); }
    F(c->att_w); F(c->ca_w1); F(c->ca_b1); F(c->ca_w2); F(c->ca_b2);
    for (int i = 0; i < 4; i++) { F(c->fcw[i]); F(c->fcb[i]); }
    F(c->act[0]); F(c->act[1]);
    c->capN = 0;
    c->loaded = false;
}

static int upload(float** dst, const std::vector<float>& v, std::string* err) {
    if (hipMalloc((void**)dst, v.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "lg_cnn_load: device allocation/copy failed";
        return LG_ERR_HIP;
    }
    return LG_OK;
}

int lg_cnn_upload(LgCnn* c, const lg_cnn_weights* w, std::string* err) {
    lg_cnn_free(c);
    const float eps = w->bn_eps > 0.f ? w->bn_eps : 1e-5f;
    // ---- layer plan: n_blocks x [conv(cin->f), conv(f->f), pool]; channels padded to the kernels' 64-channel granule
    const int nb = w->n_blocks > 0 ? w->n_blocks : 3;
    int filt[4] = {64, 128, 256, 0};
    if (w->n_blocks > 0)
        for (int b = 0; b < 4; b++) filt[b] = w->filters[b];
    if (nb < 1 || nb > 4) { *err = "lg_cnn_load: 1..4 encoder blocks"; return LG_ERR_UNSUPPORTED; }
    c->n_layers = 2 * nb;
    int cin = 9, cinp = 10, wi = 32;
    size_t per = 0;
    for (int b = 0; b < nb; b++) {
        const int f = filt[b], fp = (f + 63) / 64 * 64;
        if (f < 16 || f > 512 || (f % 16) != 0) { *err = "lg_cnn_load: encoder filters must be multiples of 16 in [16, 512]"; return LG_ERR_UNSUPPORTED; }
        c->layers[2 * b] = {cin, f, cinp, fp, wi, false};
        c->layers[2 * b + 1] = {f, f, fp, fp, wi, true};
        per = std::max(per, (size_t)fp * wi * wi);
        cin = f; cinp = fp; wi /= 2;
    }
    c->F = filt[nb - 1]; c->Fp = (c->F + 63) / 64 * 64; c->npix = wi * wi;
    c->act_per_patch = per;
    c->standard = nb == 3 && filt[0] == 64 && filt[1] == 128 && filt[2] == 256;
    if (c->layers[0].coutp != 64 && c->layers[0].coutp != 128) { *err = "lg_cnn_load: first stage wider than 128 channels"; return LG_ERR_UNSUPPORTED; }
    for (int L = 1; L < c->n_layers; L++) {
        const RtLayer& l = c->layers[L];
        if (!wino_supported(l.cinp, l.coutp, l.wi, l.pool)) {
            *err = "lg_cnn_load: encoder_filters outside the reference's configurations ([32,64,128], [64,128,256], "
                   "[64,128,256,512], [128,256,512])";
            return LG_ERR_UNSUPPORTED;
        }
    }
    for (int L = 0; L < c->n_layers; L++) {
        const RtLayer& l = c->layers[L];
        if (!w->conv_w[L] || !w->conv_b[L] || !w->bn_g[L] || !w->bn_b[L] || !w->bn_m[L] || !w->bn_v[L]) {
            *err = "lg_cnn_load: missing encoder tensor";
            return LG_ERR_INVALID;
        }
        // fold eval-mode BatchNorm2d: y = (conv + b - mean) * g / sqrt(var + eps) + beta; padded channels stay 0
        std::vector<float> bp(l.coutp, 0.0f);
        std::vector<double> scv(l.cout);
        for (int co = 0; co < l.cout; co++) {
            scv[co] = (double)w->bn_g[L][co] / sqrt((double)w->bn_v[L][co] + (double)eps);
            bp[co] = (float)(((double)w->conv_b[L][co] - (double)w->bn_m[L][co]) * scv[co] + (double)w->bn_b[L][co]);
        }
        int rc = upload(&c->bconv[L], bp, err);
        if (rc) return rc;
        if (L == 0 || c->standard) {   // direct implicit-GEMM weights [tap][cin_pad][cout] (layer 0; A/B path of the standard model)
            std::vector<float> wp((size_t)9 * l.cinp * l.coutp, 0.0f);
            for (int co = 0; co < l.cout; co++)
                for (int ci = 0; ci < l.cin; ci++)
                    for (int tap = 0; tap < 9; tap++)
                        wp[((size_t)tap * l.cinp + ci) * l.coutp + co] =
                            (float)((double)w->conv_w[L][((size_t)co * l.cin + ci) * 9 + tap] * scv[co]);
            rc = upload(&c->wconv[L], wp, err);
            if (rc) return rc;
        }
        if (L >= 1) {
            // Winograd-domain weights U = G g G^T of the BN-folded kernel, layout [ci][co][4i+j], rounded once from double
            std::vector<float> uw((size_t)l.cinp * l.coutp * 16, 0.0f);
            for (int co = 0; co < l.cout; co++) {
                for (int ci = 0; ci < l.cin; ci++) {
                    double g[3][3], gg[4][3];
                    for (int tap = 0; tap < 9; tap++)
                        g[tap / 3][tap % 3] = (double)w->conv_w[L][((size_t)co * l.cin + ci) * 9 + tap] * scv[co];
                    for (int j = 0; j < 3; j++) {
                        gg[0][j] = g[0][j];
                        gg[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
                        gg[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
                        gg[3][j] = g[2][j];
                    }
                    float* u = &uw[((size_t)ci * l.coutp + co) * 16];
                    for (int i = 0; i < 4; i++) {
                        u[4 * i + 0] = (float)gg[i][0];
                        u[4 * i + 1] = (float)(0.5 * (gg[i][0] + gg[i][1] + gg[i][2]));
                        u[4 * i + 2] = (float)(0.5 * (gg[i][0] - gg[i][1] + gg[i][2]));
                        u[4 * i + 3] = (float)gg[i][2];
                    }
                }
            }
            rc = upload(&c->uwino[L], uw, err);
            if (rc) return rc;
        }
    }
    const int F = c->F;
    c->att_type = w->attention_type;
    if (c->att_type < LG_ATT_SPATIAL || c->att_type > LG_ATT_NONE) { *err = "lg_cnn_load: unknown attention_type"; return LG_ERR_INVALID; }
    if (c->att_type == LG_ATT_SPATIAL || c->att_type == LG_ATT_HYBRID) {
        if (!w->att_w || !w->att_b) { *err = "lg_cnn_load: missing spatial attention tensor"; return LG_ERR_INVALID; }
        int rc = upload(&c->att_w, std::vector<float>(w->att_w, w->att_w + F), err);
        if (rc) return rc;
        c->att_b = w->att_b[0];
    }
    if (c->att_type == LG_ATT_CHANNEL || c->att_type == LG_ATT_HYBRID) {
        if (!w->ca_w1 || !w->ca_b1 || !w->ca_w2 || !w->ca_b2) { *err = "lg_cnn_load: missing channel attention tensor"; return LG_ERR_INVALID; }
        const int hid = F / 16;
        int rc = upload(&c->ca_w1, std::vector<float>(w->ca_w1, w->ca_w1 + (size_t)hid * F), err);
        if (!rc) rc = upload(&c->ca_b1, std::vector<float>(w->ca_b1, w->ca_b1 + hid), err);
        if (!rc) rc = upload(&c->ca_w2, std::vector<float>(w->ca_w2, w->ca_w2 + (size_t)F * hid), err);
        if (!rc) rc = upload(&c->ca_b2, std::vector<float>(w->ca_b2, w->ca_b2 + F), err);
        if (rc) return rc;
    }
    const int dims[5] = {F, F, F / 2, F / 4, 1};
    for (int L = 0; L < 4; L++) {
        const int fin = dims[L], fout = dims[L + 1];
        if (!w->fc_w[L] || !w->fc_b[L]) { *err = "lg_cnn_load: missing classifier tensor"; return LG_ERR_INVALID; }
        std::vector<float> wt((size_t)fin * fout), bt(fout);
        for (int o = 0; o < fout; o++) {
            double sc = 1.0, sh = 0.0;
            if (L < 3) {
                if (!w->fbn_g[L] || !w->fbn_b[L] || !w->fbn_m[L] || !w->fbn_v[L]) {
                    *err = "lg_cnn_load: missing classifier BN tensor";
                    return LG_ERR_INVALID;
                }
                sc = (double)w->fbn_g[L][o] / sqrt((double)w->fbn_v[L][o] + (double)eps);
                sh = (double)w->fbn_b[L][o] - (double)w->fbn_m[L][o] * sc;
            }
            bt[o] = (float)((double)w->fc_b[L][o] * sc + sh);
            for (int i = 0; i < fin; i++) wt[(size_t)i * fout + o] = (float)((double)w->fc_w[L][(size_t)o * fin + i] * sc);
        }
        int rc = upload(&c->fcw[L], wt, err);
        if (rc) return rc;
        rc = upload(&c->fcb[L], bt, err);
        if (rc) return rc;
    }
    c->loaded = true;
    return LG_OK;
}

static int lg_cnn_run_slice(LgCnn* c, const float* patches, int N, float* logits, hipStream_t s, std::string* err);

// Slices share the activation workspace (the 32-bit staging offsets of lg_wino_kernel reach 4 GiB: at most ~2 GiB of
// activations per buffer and slice, i.e. 8192 patches of the standard model).
int lg_cnn_run(LgCnn* c, const float* patches, int N, float* logits, hipStream_t s, std::string* err) {
    if (!c->loaded) { *err = "no model"; return LG_ERR_NO_MODEL; }
    const int max_slice = (int)std::max<size_t>(64, ((size_t)1 << 31) / (c->act_per_patch * sizeof(float)));
    for (int off = 0; off < N; off += max_slice) {
        const int n = N - off < max_slice ? N - off : max_slice;
        int rc = lg_cnn_run_slice(c, patches + (size_t)off * 9 * 1024, n, logits + off, s, err);
        if (rc) return rc;
    }
    return LG_OK;
}

static int lg_cnn_run_slice(LgCnn* c, const float* patches, int N, float* logits, hipStream_t s, std::string* err) {
    const size_t per = c->act_per_patch;
    const size_t tail = per + 1024;   // zeroed floats behind each buffer: >= cin * width^2 of any layer
    if (N > c->capN) {
        hipStreamSynchronize(s);
        if (c->act[0]) hipFree(c->act[0]);
        if (c->act[1]) hipFree(c->act[1]);
        c->act[0] = c->act[1] = nullptr;
        if (hipMalloc((void**)&c->act[0], ((size_t)N * per + tail) * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&c->act[1], ((size_t)N * per + tail) * sizeof(float)) != hipSuccess) {
            *err = "lg_cnn_forward: activation workspace allocation failed";
            c->capN = 0;
            return LG_ERR_NOMEM;
        }
        hipMemsetAsync(c->act[0] + (size_t)N * per, 0, tail * sizeof(float), s);
        hipMemsetAsync(c->act[1] + (size_t)N * per, 0, tail * sizeof(float), s);
        c->capN = N;
    }
    float *A = c->act[0], *B = c->act[1];
    const size_t tail_off = (size_t)c->capN * per;   // the zeroed tail sits behind BOTH buffers at the same offset
    if (c->layers[0].coutp == 64) launch_conv0<64>(patches, c, A, N, s);
    else launch_conv0<128>(patches, c, A, N, s);
    float* cur = A;
    float* nxt = B;
    if (c->standard) {
        // Winograd F(2x2,3x3) for layers 1..5 (2.25x fewer MFMA flops); LG_CNN_DIRECT=1 selects the direct implicit GEMM
        // for all layers, LG_CNN_WINO_MASK=<bits> a per-layer choice (bit L = layer L on Winograd), LG_CNN_WS_KC=8 the
        // producer/consumer form (4.45 vs 4.01 ms per 2560 patches) -- A/B and test switches, read per call.
        int wmask = getenv("LG_CNN_DIRECT") ? 0 : 0x3e;
        if (const char* e = getenv("LG_CNN_WINO_MASK")) wmask = atoi(e) & 0x3e;
        const int ws_kc = getenv("LG_CNN_WS_KC") ? atoi(getenv("LG_CNN_WS_KC")) : 0;
        static const int num_cu = [] {
            int dev = 0, n = 256;
            if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            if (const char* e = getenv("LG_CNN_WS_GRID")) n = atoi(e);
            return n > 0 ? n : 256;
        }();
#define LG_LAYER(L, KC, PP, CP)                                                                                     \
    do {                                                                                                            \
        const RtLayer& l = c->layers[L];                                                                            \
        if (!(wmask & (1 << L))) launch_conv<L, KC, PP, CP>(cur, c, nxt, N, s);                                     \
        else if (ws_kc == 8) launch_wino_ws<L, 8>(cur, c, nxt, N, s, num_cu);                                       \
        else launch_wino_rt(l.cinp, l.coutp, l.wi, l.pool, cur, c->uwino[L], c->bconv[L], nxt, cur + tail_off, N, s); \
        std::swap(cur, nxt);                                                                                        \
    } while (0)
        LG_LAYER(1, 8, 4, 1);   // 64 -> 64, pool -> 16x16
        LG_LAYER(2, 8, 4, 1);   // 64 -> 128, 16x16
        LG_LAYER(3, 8, 4, 1);   // 128 -> 128, pool -> 8x8
        LG_LAYER(4, 4, 1, 4);   // 128 -> 256, 8x8
        LG_LAYER(5, 4, 1, 4);   // 256 -> 256, pool -> 4x4
#undef LG_LAYER
    } else {
        for (int L = 1; L < c->n_layers; L++) {
            const RtLayer& l = c->layers[L];
            if (!launch_wino_rt(l.cinp, l.coutp, l.wi, l.pool, cur, c->uwino[L], c->bconv[L], nxt, cur + tail_off, N, s)) {
                *err = "lg_cnn_forward: unsupported layer shape";
                return LG_ERR_UNSUPPORTED;
            }
            std::swap(cur, nxt);
        }
    }
#define LG_HEAD(F_, NP_)                                                                                              \
    if (c->F == F_ && c->npix == NP_) {                                                                                \
        hipLaunchKernelGGL((lg_head_kernel<F_, NP_>), dim3(N), dim3(256), 0, s, cur, c->Fp, c->att_type, c->att_w, c->att_b, \
                           c->ca_w1, c->ca_b1, c->ca_w2, c->ca_b2, c->fcw[0], c->fcb[0], c->fcw[1], c->fcb[1], c->fcw[2],  \
                           c->fcb[2], c->fcw[3], c->fcb[3], logits);                                                   \
        return LG_OK;                                                                                                  \
    }
    LG_HEAD(256, 16) LG_HEAD(128, 16) LG_HEAD(512, 16) LG_HEAD(512, 4)
#undef LG_HEAD
    *err = "lg_cnn_forward: unsupported classifier size";
    return LG_ERR_UNSUPPORTED;
}
