// GraspPointCNN forward on gfx950 (see lg_cnn.h).  Reference: scripts/utils/ml_grasp_optimizer/model.py:5-128.
//
// Activation layout between the layers ("haloed planes"): a WI x WI feature map is stored as (WI + 2) rows of
// WI + 4 floats -- pixel (y, x) at [y + 1][x + 1], a zero row above and below, a zero column left, three right -- so
// that (1) the zero padding of every 3x3 convolution is DATA, not address logic, and (2) the rows a workgroup stages
// (a band of the image incl. its halo, for a chunk of channels) are contiguous 16-byte-aligned runs of global memory:
// every staging transfer is a `global_load_lds_dwordx4` (1 KiB per wave instruction).  Measured on MI355X
// (tools/ubench/mfma_issue.hip, profiles/r02_ubench_mfma_f32_issue_costs.txt): the vector-memory path moves 64 B/clk/CU
// with 16-byte lanes and 8 B/clk/CU with 4-byte lanes, whatever else the CU does -- the 4-byte staging of the first
// version of these kernels kept that path busy for 44 % of a chunk's time.  The halos are zero from allocation on (one buffer per
// layer) and stay zero: epilogues store the interior -- layer 0 also re-writes its (zero) halo bytes, so that no cache line of
// its 1.6 GB output leaves the L2 partially written (see the epilogue of lg_wino4_kernel).
//
// Conv layers: out[n][co][y][x] = relu(b[co] + sum_{ky,kx,ci} w[ky][kx][ci][co] * in[n][ci][y+ky-1][x+kx-1]).
// f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) shares the SIMD's FMA lanes with the VALU: every other vector
// instruction of a wave ADDS its 4 cycles to the MFMA time (same profile file), LDS reads do not.
#include "lg_cnn.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// lg_wino4_kernel's DMA schedule: how many of a chunk's 9 + NIN transfers (U block of the next chunk first, then the input
// rows three chunks ahead) a staging wave issues before the first MFMA group (S0) and after groups 0..3 (S1..S4; the rest after
// group 4).  The U block has to land by the end of the chunk that issues it (2-stage ring, the LDS is full): issued three after
// each of the first three groups (0 3 3 3 3) the last transfers had two thirds of a chunk to come back from L2 and the chunk's
// closing wait stalled on them -- 5.26 ms per 5120 patches; 5 4 0 0 3: 5.06-5.12; all nine after group 0 (0 9 ...): 5.51 (they
// delay the wave's own next group); all nine up front, inputs after group 3 (9 0 0 0 3): 5.04-5.06 (tools/cnn_time.py, one box).
#ifndef LG_W4_S0
#define LG_W4_S0 9
#define LG_W4_S1 0
#define LG_W4_S2 0
#define LG_W4_S3 0
#define LG_W4_S4 3
#endif
#ifndef LG_W4_NT
#define LG_W4_NT 0    // 1: non-temporal plane stores in lg_wino4_kernel's epilogue (same results)
#endif
#ifndef LG_W4_EXP
#define LG_W4_EXP 0   // timing ablations of lg_wino4_kernel (WRONG RESULTS): 1 no transform arithmetic, 2 no transform at all,
#endif                //   4 no A DMA, 8 no input DMA, 16 no fragment reads, 32 no MFMAs, 64 no epilogue, 128 no halo stores (right results),
                      //   256 epilogue arithmetic without its stores, 512 stores into one small region,
                      //   2048 / 4096 no stores from the staging / the transform waves, 8192 half the fragment reads (B operand stale),
                      //   16384 / 32768 the transform without its LDS reads / writes

namespace {

struct LayerCfg { int cin, cinp, cout, wi; bool pool; };
constexpr LayerCfg kLayers[6] = {
    {9, 10, 64, 32, false}, {64, 64, 64, 32, true},   {64, 64, 128, 16, false},
    {128, 128, 128, 16, true}, {128, 128, 256, 8, false}, {256, 256, 256, 8, true}};

constexpr int lg_wp(int wi) { return wi + 4; }                   // row pitch of a haloed plane
constexpr int lg_plane(int wi) { return (wi + 2) * (wi + 4); }   // floats per haloed plane

#define LG_DMA16(src, dst)                                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                  \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define LG_DMA4(src, dst)                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                  \
                                     (__attribute__((address_space(3))) void*)(dst), 4, 0, 0)

// [N][9][32][32] (the C-ABI's patch layout) -> the first 9 of [N][12][34][36] haloed planes; halos and planes 9..11 are never written
__global__ __launch_bounds__(256) void lg_repack_kernel(const float* __restrict__ in, float* __restrict__ out, long long nplanes) {
    const long long pl = blockIdx.x;
    if (pl >= nplanes) return;
    const float4* src = reinterpret_cast<const float4*>(in + pl * 1024);
    float* dst = out + ((pl / 9) * 12 + pl % 9) * lg_plane(32);
    const int t = threadIdx.x, y = t >> 3, x4 = (t & 7) * 4;
    const float4 v = src[t];
    float* d = dst + (y + 1) * lg_wp(32) + x4 + 1;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

// ---------------------------------------------------------------------------------------------------------
// Direct implicit GEMM, D = W * X on v_mfma_f32_32x32x2_f32: M = 32 output channels, N = 32 consecutive pixels,
// K = (tap, input channel) pairs; A (weights) and B (im2col of an LDS staged tile) one VGPR each, lanes 0..31 carry k,
// lanes 32..63 k+1.  The reference path of the tests (LG_CNN_DIRECT=1 / LG_CNN_WINO_MASK) -- not on the default path.
// WG = 256 threads = 4 waves; wave tile = 2 pixel-blocks x 2 channel-blocks; PP x CP wave arrangement.
template <int CIN, int CINP, int COUT, int WI, bool POOL, bool OUT_HALO, int KC, int PP, int CP>
__global__ __launch_bounds__(256, 2) void lg_conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         const float* __restrict__ zeros) {
    static_assert(PP * CP == 4, "4 waves per workgroup");
    constexpr int PBROWS = 32 / WI > 0 ? 32 / WI : 1;   // image rows per 32-pixel block (WI=32:1, 16:2, 8:4)
    constexpr int ROWS = 2 * PP * PBROWS;               // output rows per workgroup
    constexpr int TR = ROWS + 2;                        // staged input rows (halo 1)
    constexpr int TWID = WI + 2;
    constexpr int COUT_T = 64 * CP;
    constexpr int IN_CH_STRIDE = TR * TWID;
    constexpr int IN_ELEMS = KC * TR * TWID;
    constexpr int NIN = (IN_ELEMS + 255) / 256;
    constexpr int IN_PAD = NIN * 256;
    constexpr int W4_ELEMS = 9 * KC * (COUT_T / 4);
    constexpr int NW4 = (W4_ELEMS + 255) / 256;
    constexpr int BUF = IN_PAD + NW4 * 256 * 4;
    constexpr int PLANE = lg_plane(WI), WP = lg_wp(WI);
    __shared__ __attribute__((aligned(16))) float s_buf[2 * BUF];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wp_i = wave % PP, wc_i = wave / PP;
    constexpr int BANDS = WI / ROWS > 0 ? WI / ROWS : 1;
    const int n = blockIdx.x / (BANDS * (COUT / COUT_T));
    const int rem = blockIdx.x % (BANDS * (COUT / COUT_T));
    const int band = rem % BANDS, ct = rem / BANDS;
    const int y0 = band * ROWS;
    const int co0 = ct * COUT_T;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int p = lane & 31, kh = lane >> 5;
    int boff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int pb = wp_i * 2 + i;
        const int row = pb * PBROWS + p / WI, x = p % WI;
        boff[i] = row * TWID + x;
    }
    const int aoff = (wc_i * 64) + (lane & 31);

    const float* in_n = in + (size_t)n * CIN * PLANE;
    int off_in[NIN];   // offset inside the chunk's first haloed plane group, or -1: unused slot / padded channel
    int ci_in[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int idx = t + 256 * j;
        const int ci = idx / (TR * TWID), r2 = idx % (TR * TWID);
        const int ry = r2 / TWID, rx = r2 % TWID;
        off_in[j] = idx < IN_ELEMS ? ci * PLANE + (y0 + ry) * WP + rx : -1;   // haloed row y0 + ry = image row y0 - 1 + ry
        ci_in[j] = ci;
    }
    int off_w[NW4];
#pragma unroll
    for (int j = 0; j < NW4; j++) {
        const int idx = t + 256 * j;
        const int q = idx % (COUT_T / 4), rest = idx / (COUT_T / 4);
        const int ci = rest % KC, tap = rest / KC;
        off_w[j] = (idx < W4_ELEMS) ? (tap * CINP + ci) * COUT + co0 + 4 * q : -1;
    }
    auto issue_chunk = [&](int c0, int stage) {
        float* sb = s_buf + stage * BUF;
        const float* in_c = in_n + (size_t)c0 * PLANE;
        const float* w_c = wp + (size_t)c0 * COUT;
#pragma unroll
        for (int j = 0; j < NIN; j++) {
            const bool ok = off_in[j] >= 0 && (CIN == CINP || c0 + ci_in[j] < CIN);
            const float* src = ok ? in_c + off_in[j] : zeros;
            LG_DMA4(src, sb + 256 * j + 64 * wave);
        }
#pragma unroll
        for (int j = 0; j < NW4; j++) {
            const float* src = off_w[j] >= 0 ? w_c + off_w[j] : zeros;
            LG_DMA16(src, sb + IN_PAD + 4 * (256 * j + 64 * wave));
        }
    };
    issue_chunk(0, 0);
    int stage = 0;
    for (int c0 = 0; c0 < CINP; c0 += KC, stage ^= 1) {
        __syncthreads();
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

    // ---- epilogue: bias + ReLU (+ 2x2 max pool); interior of the haloed plane (or a dense plane for the head)
    constexpr int WO = POOL ? WI / 2 : WI;
    constexpr int OP = OUT_HALO ? lg_wp(WO) : WO;                 // output row pitch
    constexpr int OPL = OUT_HALO ? lg_plane(WO) : WO * WO;        // output plane
    constexpr int OO = OUT_HALO ? lg_wp(WO) + 1 : 0;              // offset of pixel (0, 0)
    float* out_n = out + (size_t)n * COUT * OPL + OO;
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
                    const int row = y0 + pb * PBROWS + p / WI, x = p % WI;
                    out_n[(size_t)co * OPL + row * OP + x] = fmaxf(acc[i][j][r] + bv, 0.0f);
                }
            } else {
                if (WI == 32) {
                    float v = fmaxf(acc[0][j][r], acc[1][j][r]);   // the wave's two pixel blocks are rows y0+2a, y0+2a+1
                    v = fmaxf(v, __shfl_xor(v, 1, 64));
                    const int yo = (y0 + wp_i * 2) / 2, xo = p >> 1;
                    if ((p & 1) == 0) out_n[(size_t)co * OPL + yo * OP + xo] = fmaxf(v + bv, 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        float v = acc[i][j][r];
                        v = fmaxf(v, __shfl_xor(v, WI, 64));
                        v = fmaxf(v, __shfl_xor(v, 1, 64));
                        const int pb = wp_i * 2 + i;
                        const int row = y0 + pb * PBROWS + p / WI, x = p % WI;
                        if (((p / WI) & 1) == 0 && (x & 1) == 0)
                            out_n[(size_t)co * OPL + (row >> 1) * OP + (x >> 1)] = fmaxf(v + bv, 0.0f);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Layer 0 (9 input channels, 32x32, no pool): one workgroup per (patch, 64 output channels) walks the four 8-row bands
// of its patch.  The 23 KB weight image is staged once; the next band's input (10 haloed rows x 36 floats per channel:
// 90 16-byte pieces, 4 transfers per thread and band instead of 14 four-byte ones) streams into the other LDS stage
// while the current one is multiplied.  The 10th (padding) channel reads a zeroed plane.
template <int COUT>
__global__ __launch_bounds__(256, 2) void lg_conv0_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          const float* __restrict__ zeros) {
    constexpr int CIN = 9, CINP = 10, WI = 32, ROWS = 8, TR = ROWS + 2, WP = lg_wp(WI), PLANE = lg_plane(WI), BANDS = WI / ROWS;
    constexpr int IN_CH_STRIDE = TR * WP;                 // 360 floats
    constexpr int PPC = IN_CH_STRIDE / 4;                 // 16-byte pieces per channel and band
    constexpr int NIN = (CINP * PPC + 255) / 256;
    constexpr int IN_PAD = NIN * 256 * 4;                 // floats per input stage
    constexpr int W4_ELEMS = 9 * CINP * (64 / 4);
    constexpr int NW4 = (W4_ELEMS + 255) / 256;
    __shared__ __attribute__((aligned(16))) float s_buf[2 * IN_PAD + NW4 * 256 * 4];   // [input stage 0 | stage 1 | weights]
    float* const s_w = s_buf + 2 * IN_PAD;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // pixel-block pair of this wave: band rows 2w, 2w+1
    const int n = blockIdx.x / (COUT / 64), co0 = (blockIdx.x % (COUT / 64)) * 64;
    const int p = lane & 31, kh = lane >> 5;
    const int boff0 = (2 * wave) * WP + p, boff1 = (2 * wave + 1) * WP + p;
    const int aoff = lane & 31;
    const float* in_n = in + (size_t)n * 12 * PLANE;   // 12 planes per haloed input patch (lg_cnn_halo_patch_floats)

    const float* src0[NIN];   // band 0 source of every staging slot; + ROWS * WP floats per band for real channels
    bool real[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int e = t + 256 * j;
        const int ci = e / PPC, piece = e % PPC;
        real[j] = ci < CIN;
        src0[j] = real[j] ? in_n + ci * PLANE + 4 * piece : zeros + 4 * piece;   // slots past the image: zeros, into LDS slack
    }
    auto issue_input = [&](int band, int stage) {
        float* sb = s_buf + stage * IN_PAD;
#pragma unroll
        for (int j = 0; j < NIN; j++) LG_DMA16(src0[j] + (real[j] ? band * ROWS * WP : 0), sb + 4 * (256 * j + 64 * wave));
    };
#pragma unroll
    for (int j = 0; j < NW4; j++) {   // weights once: [tap][ci][64 channels], 16 bytes per lane
        const int idx = t + 256 * j;
        const int q = idx % 16, rest = idx / 16;
        const int ci = rest % CINP, tap = rest / CINP;
        const float* src = idx < W4_ELEMS ? wp + ((size_t)tap * CINP + ci) * COUT + co0 + 4 * q : zeros;
        LG_DMA16(src, s_w + 4 * (256 * j + 64 * wave));
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
                const float b0 = s_in[ci * IN_CH_STRIDE + boff0 + ky * WP + kx];
                const float b1 = s_in[ci * IN_CH_STRIDE + boff1 + ky * WP + kx];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // bias + ReLU into the interior of the haloed output plane: 32 consecutive floats per (channel, row) and store
        float* out_n = out + (size_t)n * COUT * PLANE + WP + 1;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = co0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int y = band * ROWS + 2 * wave + i;
                    out_n[(size_t)co * PLANE + y * WP + p] = fmaxf(acc[i][j][r] + bv[j][r], 0.0f);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3):  Y = A^T [ (G g G^T) .* (B^T d B) ] A  per 4x4 input tile d / 2x2 output tile Y, 16 independent
// contractions over the input channels: 2.25x fewer MFMA flops than the direct form, still exact-f32 products and f32
// accumulation (v_mfma_f32_16x16x4_f32).  Used for the encoder shapes F(4x4,3x3) below does not cover, and as its A/B.
//
// Workgroup = 256 threads = 32 tiles x 64 output channels x all 16 Winograd positions.  Wave w owns output channels
// 16w..16w+15: M = 16 channels, N = 16 tiles, K = 4 input channels per MFMA, 16 positions x 2 tile halves = 32
// accumulators (128 VGPRs); every lane ends with all 16 positions of its (channel, tile) pairs: the output transform +
// bias + ReLU (+ the 2x2 max-pool = one output tile) needs no exchange.
//   * A operand  U: four 1 KiB-contiguous wave loads per k-step from L2 into registers, next k-step prefetched.
//   * B operand  V[ci][tile][16 (+4 pad)]: the band's haloed rows of a chunk of 8 channels stream into a 2-stage LDS ring
//     as 16-byte DMA pieces (2-3 per thread and chunk), are transformed by thread (ci, tile) = (t>>5, t&31) and read
//     back as 4 ds_read_b128 per tile half (row stride 80 B: conflict free).
template <int CIN, int COUT, int WI, bool POOL, bool OUT_HALO>
__global__ __launch_bounds__(256, 2) void lg_wino_kernel(const float* __restrict__ in, const float* __restrict__ U,
                                                         const float* __restrict__ bias, float* __restrict__ out, int N, int ntb) {
    constexpr int THREADS = 256;
    constexpr int KC = 8;                                  // input channels per chunk: one (channel, tile) item per thread
    constexpr int KS = KC / 4;                             // MFMA k-steps per chunk
    constexpr int CB = 64;                                 // output channels per workgroup
    constexpr int TC = WI / 2, TP = TC * TC;               // tile columns, tiles per patch
    constexpr int PB = TP >= 32 ? 1 : 32 / TP;             // patches per workgroup (8x8 images: 2)
    constexpr int BPP = TP >= 32 ? TP / 32 : 1;            // workgroups (row bands) per patch
    constexpr int TPB = 32 / PB;                           // tiles of one patch inside the workgroup
    constexpr int TROWS = TPB / TC;                        // tile rows per band
    constexpr int WP = lg_wp(WI), PLANE = lg_plane(WI);
    constexpr int RH = 2 * TROWS + 2;                      // staged haloed rows per channel and patch
    constexpr int RS = RH * WP, S = PB * RS;               // floats per (channel, patch) / per channel
    constexpr int PPC = RS / 4;                            // 16-byte pieces per (channel, patch)
    constexpr int NPIECE = KC * PB * PPC;
    constexpr int NIN = (NPIECE + THREADS - 1) / THREADS;
    constexpr int STAGE = NIN * THREADS * 4;               // floats per input stage
    constexpr int VS = 20;                                 // floats per (ci, tile) row of V: 16 positions + 4 pad
    constexpr int NCB = COUT / CB;
    constexpr int NC = CIN / KC;
    static_assert(CIN % KC == 0 && COUT % CB == 0 && TPB % TC == 0 && KS % 2 == 0 && (RS % 4) == 0, "shape");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float s_mem[2 * STAGE + KC * 32 * VS];
    float* const s_v = s_mem + 2 * STAGE;

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
    const int y0 = band * 2 * TROWS;                       // first output row of the band = first staged haloed row
    const int co0 = cb * CB;

    // ---- input staging: chunk-invariant byte offsets of this thread's 16-byte pieces from a base that advances per chunk
    const float* in_n = in + (size_t)n0 * CIN * PLANE;
    unsigned voff[NIN];
#pragma unroll
    for (int j = 0; j < NIN; j++) {
        const int e = t + THREADS * j;
        const int ci = e / (PB * PPC), r = e % (PB * PPC);
        const int pb = r / PPC, piece = r % PPC;
        const bool ok = e < NPIECE && n0 + pb < N;         // slots past the image / patches past the batch: any valid bytes
        voff[j] = ok ? 4u * (unsigned)((pb * CIN + ci) * PLANE + y0 * WP + 4 * piece) : 0u;
    }
    auto issue_input = [&](int c, int stage) {
        const char* in_c = (const char*)(in_n + (size_t)c * KC * PLANE);
        float* sb = s_mem + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NIN; j++) LG_DMA16(in_c + voff[j], sb + 4 * (THREADS * j + 64 * wave));
    };
    // ---- A operand: lane (co = lane & 15, k = lane >> 4) reads the 16 positions of U[ci][co], stored
    //      [k-step][16-channel block][position group q][lane][4 positions]: every wave instruction reads 1 KiB of CONSECUTIVE
    //      bytes (16 cache lines).  The first layout, [ci][co][16], had each lane read its own 64-byte row: 64 lines per
    //      instruction, and the vector L1 looks up one line per clock -- 8 such loads per wave and chunk kept it busy for
    //      the whole chunk (0.60 of the MFMA peak with everything else in place).
    const float* u_lane = U + ((size_t)(co0 / 16 + wave) * 256 + lane) * 4;
    auto load_u = [&](int ks, f32x4 (&a)[4]) {             // ks = global k-step (4 input channels each)
        const float* p = u_lane + (size_t)ks * (COUT / 16) * 1024;
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = *reinterpret_cast<const f32x4*>(p + 256 * q);
    };
    // ---- transform role: (channel of the chunk, tile)
    const int tci = t >> 5, tau = t & 31;
    const int tpb = tau / TPB, ttl = tau % TPB;
    const int ttr = ttl / TC, ttc = ttl % TC;
    const int tsrc = tci * S + tpb * RS + (2 * ttr) * WP + 2 * ttc;
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
        {   // V = B^T d B for (tci, tau): plain v_add / v_sub (a packed add costs 2x a plain one beside f32 MFMA: no gain)
            const float* sp = s_mem + (c & 1) * STAGE + tsrc;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // plain float reads (merged into ds_read2_b64): a float2-typed read makes hipcc drain vmcnt, i.e. the
                // next chunk's DMA prefetch, in front of it
                d[i][0] = sp[i * WP]; d[i][1] = sp[i * WP + 1]; d[i][2] = sp[i * WP + 2]; d[i][3] = sp[i * WP + 3];
            }
            float r[4][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                r[0][j] = d[0][j] - d[2][j];
                r[1][j] = d[1][j] + d[2][j];
                r[2][j] = d[2][j] - d[1][j];
                r[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 v = {r[i][0] - r[i][2], r[i][1] + r[i][2], r[i][2] - r[i][1], r[i][1] - r[i][3]};
                *reinterpret_cast<f32x4*>(tdst + 4 * i) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // V visible; does not drain the prefetches
#pragma unroll
        for (int ksl = 0; ksl < KS; ksl++)
            mfma_step(ksl, ab[ksl & 1], [&] {
                if (ksl >= 1) {
                    const int ksn = KS * c + ksl + 1;
                    load_u(ksn < KS * NC ? ksn : KS * NC - 1, ab[(ksl + 1) & 1]);
                }
            });
    }

    // ---- epilogue: Y = A^T M A, bias, ReLU (+ max over the 2x2 tile); interior of the haloed output plane
    constexpr int WO = POOL ? WI / 2 : WI;
    constexpr int OP = OUT_HALO ? lg_wp(WO) : WO;
    constexpr int OPL = OUT_HALO ? lg_plane(WO) : WO * WO;
    constexpr int OO = OUT_HALO ? lg_wp(WO) + 1 : 0;
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
                float* o = out + ((size_t)n * COUT + co) * OPL + OO;
                if (POOL) {
                    o[trg * OP + tcg] = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)), 0.0f);
                } else {
                    float* o0 = o + (2 * trg) * OP + 2 * tcg;    // (interior pixels start at an odd column: dword stores)
                    o0[0] = fmaxf(y00, 0.f); o0[1] = fmaxf(y01, 0.f);
                    o0[OP] = fmaxf(y10, 0.f); o0[OP + 1] = fmaxf(y11, 0.f);
                }
            }
        }
    }
}

// LDS fragment reads hipcc does not see (it drains lgkmcnt to 0 in front of every MFMA group of this kernel instead of counting):
// the read and, later, a wait that names the destinations (so no consumer is scheduled above it) with K = the number of
// THESE reads issued after the one waited for.  LDS operations complete in order, so compiler-issued LDS operations in
// between only make the wait longer, never too short.
typedef float lg_f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lg_lds_read16(lg_f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int K>
__device__ __forceinline__ void lg_lds_wait2(lg_f32x4& a, lg_f32x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(K) : "memory");
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(4x4,3x3):  Y = A^T [ (G g G^T) .* (B^T d B) ] A  per 6x6 input tile d / 4x4 output tile Y: 36 independent
// contractions over the input channels, 4x fewer MFMA flops than the direct form (F(2x2,3x3): 2.25x).  Products and
// accumulation stay exact f32 (v_mfma_f32_16x16x4_f32); the transforms' small integer / dyadic coefficients cost ~1e-6 of
// relative error on the logits (tests compare with the reference's fp64 logits at 1e-4; the F(2x2,3x3) and direct kernels
// remain selectable as second opinions: LG_CNN_F23=1, LG_CNN_DIRECT=1).
//
// Work item = 64 output channels x 32 tiles x all 36 positions, in chunks of 4 input channels (one MFMA k-step).
// Workgroup = 512 threads = 8 waves, PERSISTENT: one per CU, walking its items as ONE software pipeline over
// (item, chunk) -- the staging of an item's first chunks runs under the previous item's last MFMAs.  Wave w owns channel
// block cb = w & 3 (16 channels) and tile block tb = w >> 2 (16 tiles): 36 accumulators (144 VGPRs); a lane ends with all
// 36 positions of its 4 (channel, tile) pairs, so the output transform, bias, ReLU and the 2x2 max-pool need no exchange.
//   * A operand: the chunk's 36 KB block of U -- stored in global memory in exactly the order the fragments are read,
//     [k-step][64-channel block][cb][position group of 4][lane = (k, channel)][4 positions] -- streams into a 2-stage LDS ring
//     by 16-byte DMA (4.5 transfers per thread and chunk) and is shared by the two tile-block waves of a channel block:
//     one ds_read_b128 per 4 MFMAs.
//   * B operand: the band's haloed input rows of the chunk's 4 channels arrive by 16-byte DMA (2-3 transfers per thread),
//     are transformed (V = B^T d B, work item = (channel, tile, row of V): 768 items per chunk on 512 threads) into
//     V[tb][position group][lane = (k, tile)][4 positions] -- again one ds_read_b128 per 4 MFMAs, shared by 4 waves.
//   Inside a chunk the fragment reads run two position groups ahead of the MFMAs that use them (3-deep register ring,
//   hand-counted lgkmcnt waits); the work for the NEXT chunk is split by role between the two waves of every SIMD (see
//   "roles" in the kernel): one transforms, the other issues the DMA transfers; ONE barrier per chunk.
template <int CIN, int COUT, int WI, bool POOL, bool OUT_HALO, bool COB_MAJOR, bool SPLIT>
__global__ __launch_bounds__(512) void lg_wino4_kernel(const float* __restrict__ in, const float* __restrict__ U4,
                                                       const float* __restrict__ bias, float* __restrict__ out, int N, int ntb,
                                                       float* __restrict__ kpart, unsigned* __restrict__ kflag, unsigned* __restrict__ kerr) {
    constexpr int KC = 4;
    constexpr int TC = WI / 4, TP = TC * TC;               // tile columns, tiles per patch
    constexpr int PB = TP >= 32 ? 1 : 32 / TP;             // patches per item
    constexpr int BPP = TP >= 32 ? TP / 32 : 1;            // items (row bands) per patch
    constexpr int TPB = 32 / PB;                           // tiles of one patch inside the item
    constexpr int TROWS = TPB / TC;                        // tile rows per band
    constexpr int WP = lg_wp(WI), PLANE = lg_plane(WI);
    constexpr int RH = 4 * TROWS + 2;                      // staged haloed rows per channel and patch
    // LDS image of the staged rows.  Row pitch WPL: the transform reads a lane's tile rows as 16-byte pieces, lane = (channel
    // bit, tile), and with the planes' own pitch (WI + 4) every such read took 8 LDS cycles instead of 4 -- bank conflicts, 90 % of
    // the kernel's conflict cycles (tools/ubench/lds_conflict.hip, profiles/r04_ubench_lds_conflicts.txt: which lane -> address
    // patterns a ds_read_b128 takes at full rate is not the textbook rule; measured).  32 x 32 images: pitch 40, 16 x 16: 24 are
    // conflict-free; 8 x 8 keeps pitch 12 and deals the wave's 8-lane groups to the tiles in the order 0 1 3 2 (tperm below).
    // The pad pieces of a row are slots of the DMA enumeration that fetch a harmless valid address.
    constexpr int WPL = WI == 32 ? 40 : WI == 16 ? 24 : WP;
    constexpr int RS = RH * WPL, S = PB * RS;
    constexpr int PPC = RS / 4;
    constexpr int RPC = WPL / 4, RPV = WP / 4;             // 16-byte slots per LDS row, of which the first RPV hold data
    constexpr int NPIECE = KC * PB * PPC;
    constexpr int NIN = (NPIECE + 255) / 256;              // input transfers per chunk and thread of the four staging waves
    constexpr int STAGE = NIN * 256 * 4;                   // floats per input stage
    constexpr int ABLK = 64 * 36 * KC;                     // 9216 floats of U per (k-step, 64-channel block)
    constexpr int VBLK = 32 * 36 * KC;                     // 4608 floats of V per chunk
    constexpr int NCB = COUT / 64;
    constexpr int NC = CIN / KC;
    static_assert(CIN % KC == 0 && COUT % 64 == 0 && TPB % TC == 0 && (RS % 4) == 0 && TROWS >= 1, "shape");
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // Input ring: 3 stages where they fit in the 160 KB (every shape but the 4x4 images of the deepest encoder variant).  The
    // input rows come from HBM (the previous layer's output, GBs per layer): with 3 stages the transfers of chunk q + 3 are
    // issued during chunk q and only have to land by the END of chunk q + 1 (a counted vmcnt wait leaves them in flight across
    // one barrier); with 2 stages they have to land within the chunk that issues them and every chunk pays an HBM round trip.
    constexpr int NSTG = (2 * ABLK + 2 * VBLK + 3 * STAGE + COUT) * 4 <= 160 * 1024 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float s_mem[2 * ABLK + 2 * VBLK + NSTG * STAGE + COUT];
    float* const s_a = s_mem;
    float* const s_v = s_mem + 2 * ABLK;
    float* const s_in = s_mem + 2 * ABLK + 2 * VBLK;
    float* const s_bias = s_in + NSTG * STAGE;             // the layer's (BN-folded) biases

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // XCD-aware item order (workgroups b, b + 8, ... share an XCD and its L2).  XCD x owns the tile blocks 8 * local + x and all
    // NCB channel blocks of them, as a list of ntbx * NCB items; its gx workgroups take list positions j, j + gx, j + 2 gx, ...
    // so at any time they work on neighbouring positions.  Small U (fits L2 beside the inputs): channel block fastest, the NCB
    // readers of an input band run back to back.  Large U (layers 4-5: 4.7 / 9.4 MB against 4 MB of L2): channel block
    // slowest, every CU of the XCD streams the SAME 64-channel slice of U while the XCD's input bands stream past.
    //
    // The tail: llen items on gx workgroups are R full rounds and r left-over items.  Left over, they would cost a whole round
    // on r of the gx workgroups (640 patches: 40 items of an 8 x 8 layer on 32 workgroups per XCD -- two rounds where 1.25 would
    // do; one frame: 4 items, 28 workgroups idle).  Each left-over item is split over P workgroups along the INPUT CHANNELS
    // (chunks [part * NC / P, (part + 1) * NC / P): the chunk pipeline is unchanged, it only starts and ends elsewhere);
    // the first P - 1 parts publish their partial outputs through L2 (kpart, 128 KB each) and count themselves in (kflag);
    // the last part -- the highest workgroup index of the P, dispatched after the others -- waits for the count, adds the
    // partial outputs and stores the item.  A split item is always its workgroup's last.
    const int ntbx = (ntb + 7) / 8;
    const int xcd = blockIdx.x % 8, jx = blockIdx.x / 8, gx = gridDim.x / 8;
    const int llen = ntbx * NCB;
    const int R = llen / gx, rem = llen - R * gx;
    int P = 1;
    if (SPLIT && rem > 0) {   // (SPLIT = false: the launcher found no split worth its exchange; the left-over items run whole)
        const int cap = min(min(gx / rem, 8), NC / 4);
        while (2 * P <= cap) P *= 2;
    }
    const bool has_tail = jx < rem * P;
    const int ni = R + (has_tail ? 1 : 0);                          // items of this workgroup
    if (ni == 0) return;
    const int CPT = NC / P;                                         // chunks of a part
    const int Q = R * NC + (has_tail ? CPT : 0);                    // chunks of this workgroup, all items

    struct Item { int cob, n0, band, c0, c1, part; };
    auto item_at = [&](int i) {
        const bool last = i == R;                                   // the (possibly split) left-over item
        const int p = last ? R * gx + jx / P : jx + i * gx;
        Item it;
        it.part = last ? jx % P : 0;
        it.c0 = last ? it.part * CPT : 0;
        it.c1 = last ? it.c0 + CPT : NC;
        it.cob = COB_MAJOR ? p / ntbx : p % NCB;
        const int tbk = (COB_MAJOR ? p % ntbx : p / NCB) * 8 + xcd;   // tile blocks past ntb: patches >= N, nothing is stored
        it.n0 = PB > 1 ? tbk * PB : tbk / BPP;
        it.band = PB > 1 ? 0 : tbk % BPP;
        return it;
    };

    // MFMA roles: channel block cb, tile block tbw; fragment addresses
    const int cb = wave & 3, tbw = wave >> 2;
    const float* const a_rd = s_a + ((cb * 9) * 64 + lane) * 4;
    const float* const b_rd = s_v + ((tbw * 9) * 64 + lane) * 4;

    // ================================================================================================ roles
    // Waves w and w + 4 share a SIMD.  Both run the same MFMA sequence on their own (channel block, tile block); beside it
    //   waves 0..3 TRANSFORM: wave w turns channel k = 2 (w >> 1) + (lane >> 5) of the NEXT chunk into rows 3 (w & 1) ..
    //              3 (w & 1) + 2 of V for tile lane & 31 (the half is wave-uniform: straight-line code per half, the row
    //              stage computed only for the rows the half needs: 72 VALU per thread and chunk, no redundancy);
    //   waves 4..7 STAGE:     they issue every LDS-DMA transfer (U block of the next chunk, input rows of the chunk after),
    //              a few after each of the first MFMA groups, so the transfers' issue time hides behind matrix work and the
    //              last one has half a chunk to land.
    // The two roles are two copies of the whole item loop (accumulators never cross a role join); both meet at the same
    // barriers: two in the prologue, one per chunk.
    auto role = [&](auto is_t_tag) {
        constexpr bool IS_T = decltype(is_t_tag)::value;
        const int dw = wave - 4;                                // staging wave index (IS_T == false)

        // ---- staging state (waves 4..7)
        const float* in_item = in;
        unsigned voff[NIN];
        int in_i = 0, in_c = 0, in_cend = NC;                   // (item, chunk) the input stream issues next; the item's last chunk + 1
        auto set_input_item = [&](int i) {
            const Item it = item_at(i);
            in_c = it.c0;
            in_cend = it.c1;
            const int nb = it.n0 < N ? it.n0 : 0;               // addressing stays inside the buffer for discarded items
            in_item = in + (size_t)nb * CIN * PLANE;
            const int y0 = it.band * 4 * TROWS;                  // first staged haloed row (= output row y0 - 1)
#pragma unroll
            for (int j = 0; j < NIN; j++) {
                const int e = (j * 4 + dw) * 64 + lane;
                const int ci = e / (PB * PPC), r = e % (PB * PPC);
                const int pb = r / PPC, piece = r % PPC;
                const int prow = piece / RPC, pcol = piece % RPC;
                const bool ok = e < NPIECE && nb + pb < N && pcol < RPV;
                voff[j] = ok ? 4u * (unsigned)((pb * CIN + ci) * PLANE + (y0 + prow) * WP + 4 * pcol) : 0u;
            }
        };
        // transfer j of the input chunk the stream points at (then the stream advances after the last one)
        auto input_piece = [&](int j, int stage) {
            if (LG_W4_EXP & 8) return;
            const char* in_c_ptr = (const char*)(in_item + (size_t)in_c * KC * PLANE);
            LG_DMA16(in_c_ptr + voff[j], s_in + stage * STAGE + 4 * ((j * 4 + dw) * 64));
        };
        auto input_advance = [&]() {
            if (++in_c == in_cend) {
                if (++in_i < ni) set_input_item(in_i);
            }
        };
        // transfer j (0..8) of the A block of (cob, chunk c): 2304 16-byte pieces = 9 x 4 waves x 64 lanes
        const unsigned a_lane_off = 16u * (unsigned)(dw * 64 + lane);   // uniform base + one 32-bit lane offset: no address VALU per transfer
        auto a_piece = [&](int j, const float* ablk, int stage) {
            if (LG_W4_EXP & 4) return;
            const char* base = (const char*)(ablk + 4 * (j * 4 * 64));
            asm("" : "+s"(base));   // (opaque: otherwise hipcc folds the lane offset into one 64-bit VGPR base and adds j * 4 KiB with a VALU op per transfer)
            LG_DMA16(base + a_lane_off, s_a + stage * ABLK + 4 * ((j * 4 + dw) * 64));
        };

        // ---- transform state (waves 0..3): channel tk, tile tau, half th (rows 3 th .. 3 th + 2 of V)
        const int th = wave & 1;
        // (8 x 8 images: the 8-lane groups 2 and 3 of each half-wave swap their tiles -- see WPL above)
        const int tk = ((wave >> 1) << 1) | (lane >> 5), tau = WI == 8 ? ((lane & 31) ^ ((lane & 16) >> 1)) : (lane & 31);
        const int tsrc = tk * S + (tau / TPB) * RS + (4 * ((tau % TPB) / TC)) * WPL + 4 * ((tau % TPB) % TC) + th * WPL;  // half 1 starts at row 1
        const int tdst = (((tau >> 4) * 9) * 64 + (tk * 16 + (tau & 15))) * 4;
        // rows th .. th + 4 of the 6x6 tile (half 0 needs rows 0..4, half 1 rows 1..5)
        auto transform_load = [&](int stage, f32x4 (&dl)[5], f32x4 (&dh)[5]) {
            if (LG_W4_EXP & 2) return;
            if (LG_W4_EXP & 16384) {   // ablation: the transform without its LDS reads
#pragma unroll
                for (int q = 0; q < 5; q++) { dl[q] = (f32x4){1.f, 2.f, 3.f, (float)lane}; dh[q] = (f32x4){4.f, (float)stage, 0.f, 0.f}; asm volatile("" : "+v"(dl[q]), "+v"(dh[q])); }
                return;
            }
            const float* sp = s_in + stage * STAGE + tsrc;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                dl[q] = *reinterpret_cast<const f32x4*>(sp + q * WPL);      // 16-byte aligned: tile columns start at multiples of 4
                // columns 4..5 of the tile, read as 16 bytes (columns 6..7 lie inside the row: WP = WI + 4): an 8-byte read with a
                // 16-byte lane stride takes the same 4 LDS cycles and counts 2 of them as bank conflict
                dh[q] = *reinterpret_cast<const f32x4*>(sp + q * WPL + 4);
            }
        };
        auto col6 = [](const float (&r)[6], float (&v)[6]) {   // 1-D transform B^T of a 6-vector (12 VALU)
            v[0] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
            const float a = fmaf(-4.f, r[2], r[4]), b = fmaf(-4.f, r[1], r[3]);
            v[1] = a + b; v[2] = a - b;
            const float cc = r[4] - r[2], e = r[3] - r[1];
            v[3] = fmaf(2.f, e, cc); v[4] = fmaf(-2.f, e, cc);
            v[5] = fmaf(4.f, r[1], fmaf(-5.f, r[3], r[5]));
        };
        auto transform_store = [&](int vstage, const f32x4 (&dl)[5], f32x4 (&dh)[5]) {
            if (LG_W4_EXP & 2) return;
            // (all four floats of dh count as used: hipcc otherwise narrows those loads to ds_read2_b64 / ds_read_b64 -- the 8-byte
            //  reads whose bank conflicts the 16-byte form avoids; here, where the values are consumed, not at the loads)
#pragma unroll
            for (int q = 0; q < 5; q++) asm("" : "+v"(dh[q]));
            float* g0 = s_v + vstage * VBLK + tdst;             // lane (k, tau & 15) of tile block tau >> 4; position p at (p >> 2) * 256 + (p & 3)
            auto D = [&](int q, int j) { return j < 4 ? dl[q][j & 3] : dh[q][j & 1]; };
            float r[3][6];
            if (th == 0) {      // rows 0,1,2 of B^T d from tile rows 0..4 (= loaded rows 0..4)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    r[0][j] = fmaf(4.f, D(0, j), fmaf(-5.f, D(2, j), D(4, j)));
                    const float a = fmaf(-4.f, D(2, j), D(4, j)), b = fmaf(-4.f, D(1, j), D(3, j));
                    r[1][j] = a + b; r[2][j] = a - b;
                }
            } else {            // rows 3,4,5 from tile rows 1..5 (= loaded rows 0..4)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const float cc = D(3, j) - D(1, j), e = D(2, j) - D(0, j);
                    r[0][j] = fmaf(2.f, e, cc); r[1][j] = fmaf(-2.f, e, cc);
                    r[2][j] = fmaf(4.f, D(0, j), fmaf(-5.f, D(2, j), D(4, j)));
                }
            }
            float v[3][6];
#pragma unroll
            for (int i = 0; i < 3; i++) col6(r[i], v[i]);
            if (LG_W4_EXP & 1) { v[0][0] = D(0, 0); v[1][1] = D(1, 1); v[2][2] = D(2, 2); }
            // hipcc's SLP vectoriser seeds on the vector stores below and packs the column stage into v_pk_fma_f32 plus dozens of
            // register moves per chunk; packed f32 issues at half rate here (tools/ubench/mfma_issue.hip), so the moves are pure
            // cost on a SIMD whose every vector instruction adds to the MFMA time.  An opaque (empty, non-volatile) asm per value
            // ends the vectoriser's use-def walk at the store operands: 5.24 -> 5.13 ms per 5120 patches (the row stage stays
            // packed: its operand pairs are adjacent registers of the 16-byte loads, no moves).  Instantiating column stage and
            // stores per half as well (no moves left in front of the stores) was SLOWER: 5.13 vs 4.96 ms on one box.
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 6; j++) asm("" : "+v"(v[i][j]));
            if (LG_W4_EXP & 32768) {   // ablation: the transform without its LDS writes
#pragma unroll
                for (int i = 0; i < 3; i++) asm volatile("" ::"v"(v[i][0]), "v"(v[i][1]), "v"(v[i][2]), "v"(v[i][3]), "v"(v[i][4]), "v"(v[i][5]));
                return;
            }
            if (th == 0) {      // positions 0..17
                *reinterpret_cast<f32x4*>(g0) = (f32x4){v[0][0], v[0][1], v[0][2], v[0][3]};
                *reinterpret_cast<f32x4*>(g0 + 256) = (f32x4){v[0][4], v[0][5], v[1][0], v[1][1]};
                *reinterpret_cast<f32x4*>(g0 + 512) = (f32x4){v[1][2], v[1][3], v[1][4], v[1][5]};
                *reinterpret_cast<f32x4*>(g0 + 768) = (f32x4){v[2][0], v[2][1], v[2][2], v[2][3]};
                *reinterpret_cast<f32x2*>(g0 + 1024) = (f32x2){v[2][4], v[2][5]};
            } else {            // positions 18..35
                *reinterpret_cast<f32x2*>(g0 + 1024 + 2) = (f32x2){v[0][0], v[0][1]};
                *reinterpret_cast<f32x4*>(g0 + 1280) = (f32x4){v[0][2], v[0][3], v[0][4], v[0][5]};
                *reinterpret_cast<f32x4*>(g0 + 1536) = (f32x4){v[1][0], v[1][1], v[1][2], v[1][3]};
                *reinterpret_cast<f32x4*>(g0 + 1792) = (f32x4){v[1][4], v[1][5], v[2][0], v[2][1]};
                *reinterpret_cast<f32x4*>(g0 + 2048) = (f32x4){v[2][2], v[2][3], v[2][4], v[2][5]};
            }
        };

        // ---- prologue: chunk 0 staged and transformed, the U block of chunk 1 and the inputs of chunks 1, 2 under way
        if (!IS_T) {
            set_input_item(0);
#pragma unroll
            for (int j = 0; j < NIN; j++) input_piece(j, 0);
            input_advance();
            const float* ablk0 = U4 + (size_t)item_at(0).cob * ABLK + (size_t)item_at(0).c0 * NCB * ABLK;
#pragma unroll
            for (int j = 0; j < 9; j++) a_piece(j, ablk0, 0);
        }
        for (int i = t; i < COUT; i += 512) s_bias[i] = bias[i];   // (read back at every item start: LDS, no vmcnt involved)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (IS_T) {
            f32x4 dl[5];
            f32x4 dh[5];
            transform_load(0, dl, dh);
            transform_store(0, dl, dh);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (Q > 1) {
#pragma unroll
                for (int j = 0; j < NIN; j++) input_piece(j, 1);
                input_advance();
                const float* ablk1 = U4 + (size_t)item_at(0).cob * ABLK + (size_t)(item_at(0).c0 + 1) * NCB * ABLK;   // (an item has >= 3 chunks: chunk 1 is item 0's)
#pragma unroll
                for (int j = 0; j < 9; j++) a_piece(j, ablk1, 1);
            }
            if (NSTG == 3 && Q > 2) {   // chunk 2 may stay in flight across the barrier
#pragma unroll
                for (int j = 0; j < NIN; j++) input_piece(j, 2);
                input_advance();
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIN) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_barrier" ::: "memory");

        // ---- main pipeline.  One iteration = one chunk q (A / V stage st = q & 1):
        //        [T: load the input rows of chunk q + 1]  MM 0..1  [T: transform -> V[st ^ 1]]  MM 2..3  [S: input DMA of chunk
        //        q + 3]  MM 4..5, fragment reads of groups 6..8 -> registers, WAIT, BARRIER(q), [S: U block of chunk q + 2 -> A[st]]
        //        MM 6..8 interleaved with the fragment reads of groups 0..2 of chunk q + 1.
        //      The barrier sits INSIDE the chunk's MFMA sequence: when it falls every wave holds the operands of three more MFMA
        //      groups in registers, so neither the arrival skew of the eight waves nor the LDS latency of the next chunk's first
        //      fragments leaves the matrix pipe idle (at a chunk boundary all eight waves asked for their first fragments at once
        //      and waited for them: tools/cnn_layers.sh, r3).  What barrier(q) promises: every read of A[st] / V[st] has returned
        //      (lgkmcnt(0) of every wave), V[st ^ 1] is written, U(q + 1) and the inputs of chunk q + 2 have landed.
        //      Stores: a wave's plane stores at an item's end are younger than the U block it asked for right after the barrier
        //      before them, so the staging waves' wait at the next barrier can leave them in flight (vmcnt counts in order; on
        //      this part loads and stores share the counter) -- a store's round trip under a write-heavy layer is longer than a chunk.
        constexpr int SCNT = 4 * (POOL ? 2 : 4);                // plane-store instructions of one epilogue that are always issued
        int q = 0;                                              // chunk counter over all items (stage parity)
        f32x4 fa[3] = {}, fb[3] = {};
#define LG_FRAG_AT(PG, AP, BP) if (!(LG_W4_EXP & 16)) { lg_lds_read16<(PG) * 1024>(fa[(PG) % 3], AP); if (!(LG_W4_EXP & 8192)) lg_lds_read16<(PG) * 1024>(fb[(PG) % 3], BP); }
        {
            const unsigned ap0 = (unsigned)(size_t)a_rd, bp0 = (unsigned)(size_t)b_rd;
            LG_FRAG_AT(0, ap0, bp0); LG_FRAG_AT(1, ap0, bp0); LG_FRAG_AT(2, ap0, bp0);
        }
        bool prev_full = false;                                 // the previous item's epilogue issued its SCNT stores (wave-uniform)
        f32x4 acc[36];
#pragma unroll
        for (int p = 0; p < 36; p++) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (item_at(0).part == 0)   // (the bias enters once: through the part that holds the first chunks)
            acc[7] = *reinterpret_cast<const f32x4*>(&s_bias[item_at(0).cob * 64 + 16 * cb + 4 * (lane >> 4)]);
#pragma unroll 1
        for (int it_i = 0; it_i < ni; it_i++) {
            const Item cur = item_at(it_i);
            const Item nxt = it_i + 1 < ni ? item_at(it_i + 1) : cur;
            const int cob_next = nxt.cob;
#pragma unroll 1
            for (int c = cur.c0; c < cur.c1; c++, q++) {
                const int st = q & 1;                           // A / V stage of this chunk
                const int in_next = NSTG == 3 ? (q + 1) % 3 : st ^ 1;   // input stage holding chunk q + 1 (transformed now)
                const int in_free = NSTG == 3 ? q % 3 : st;             // input stage the staging waves refill (chunk q + NSTG)
                // Fragment ring: position group pg uses slot pg % 3, its reads are issued two groups ahead of its MFMAs (groups 0..2
                // in the previous iteration); every wait leaves the (up to) 4 fragment reads issued after the awaited pair in flight.
                const unsigned ap = (unsigned)(size_t)(a_rd + st * ABLK), bp = (unsigned)(size_t)(b_rd + st * VBLK);
                const unsigned apn = (unsigned)(size_t)(a_rd + (st ^ 1) * ABLK), bpn = (unsigned)(size_t)(b_rd + (st ^ 1) * VBLK);
                const bool has_next = q + 1 < Q;
#define LG_FRAG(PG) LG_FRAG_AT(PG, ap, bp)
#define LG_MM(PG)                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if (!(LG_W4_EXP & 32)) _Pragma("unroll") for (int e = 0; e < 4; e++)                                                 \
        acc[4 * (PG) + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[(PG) % 3][e], fb[(PG) % 3][e], acc[4 * (PG) + e], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)   /* hipcc moves register-only instructions across asm statements, the barrier included */
                if (IS_T) {
                    f32x4 dl[5];
                    f32x4 dh[5];
                    transform_load(in_next, dl, dh);             // (after the very last chunk: stale bytes into an unused V stage)
                    lg_lds_wait2<4>(fa[0], fb[0]); LG_MM(0);
                    LG_FRAG(3);
                    lg_lds_wait2<4>(fa[1], fb[1]); LG_MM(1);
                    transform_store(st ^ 1, dl, dh);             // hipcc drains lgkmcnt here: the loads' latency has passed
                    __builtin_amdgcn_sched_barrier(0);
                    LG_FRAG(4);
                    lg_lds_wait2<4>(fa[2], fb[2]); LG_MM(2);
                    LG_FRAG(5);
                    lg_lds_wait2<4>(fa[0], fb[0]); LG_MM(3);
                    LG_FRAG(6);
                    lg_lds_wait2<4>(fa[1], fb[1]); LG_MM(4);
                    LG_FRAG(7);
                    lg_lds_wait2<4>(fa[2], fb[2]); LG_MM(5);
                    LG_FRAG(8);
                    // (no vmcnt: these waves issue plane stores only, nothing they wait for)
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]), "+v"(fa[2]), "+v"(fb[2])::"memory");
                } else {
                    const bool more2 = q + NSTG < Q;
                    lg_lds_wait2<4>(fa[0], fb[0]); LG_MM(0);
                    LG_FRAG(3);
                    lg_lds_wait2<4>(fa[1], fb[1]); LG_MM(1);
                    LG_FRAG(4);
                    lg_lds_wait2<4>(fa[2], fb[2]); LG_MM(2);
                    LG_FRAG(5);
                    lg_lds_wait2<4>(fa[0], fb[0]); LG_MM(3);
                    if (more2) {                                 // input rows NSTG chunks ahead: the stage the transform of chunk q read
#pragma unroll
                        for (int k = 0; k < NIN; k++) input_piece(k, in_free);
                        input_advance();
                    }
                    LG_FRAG(6);
                    lg_lds_wait2<4>(fa[1], fb[1]); LG_MM(4);
                    LG_FRAG(7);
                    lg_lds_wait2<4>(fa[2], fb[2]); LG_MM(5);
                    LG_FRAG(8);
                    // U(q + 1) (asked for right after the previous barrier) and everything older have landed; what may stay in
                    // flight is younger: the NIN input transfers above (3-stage ring) and, in an item's first chunk, the previous
                    // item's plane stores
                    const bool keep_in = NSTG == 3 && more2, keep_st = NSTG == 3 && c == cur.c0 && prev_full;
                    // (the vmcnt waits carry no register operands: tying the fragment registers to four alternative asm statements made
                    //  hipcc copy them -- destinations of LDS reads still in flight -- in front of the wait)
                    if (keep_in && keep_st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SCNT + NIN) : "memory");
                    else if (keep_st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SCNT) : "memory");
                    else if (keep_in) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIN) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]), "+v"(fa[2]), "+v"(fb[2])::"memory");
                    if (q + 2 < Q) {                             // the U block two chunks ahead into the stage this chunk has just released
                        const bool same = c + 2 < cur.c1;
                        const float* ablk = U4 + (size_t)(same ? cur.cob : cob_next) * ABLK + (size_t)(same ? c + 2 : c + 2 - cur.c1 + nxt.c0) * NCB * ABLK;
#pragma unroll
                        for (int k = 0; k < 9; k++) a_piece(k, ablk, st);
                    }
                }
                LG_MM(6);
                if (has_next) { LG_FRAG_AT(0, apn, bpn); }
                LG_MM(7);
                if (has_next) { LG_FRAG_AT(1, apn, bpn); }
                LG_MM(8);
                if (has_next) { LG_FRAG_AT(2, apn, bpn); }
#undef LG_FRAG
#undef LG_MM
            }

            // ---- item done: Y = A^T M A (4x4 from 6x6), ReLU (+ 2x2 max-pool); interior of the haloed output plane.  The bias is
            //      already in: column 1 of A^T is all ones, so a constant b in M[1][1] adds b to each of the 16 outputs -- the
            //      accumulator of position (1,1) starts from the bias instead of 0 (one LDS read per item instead of 64 adds).
            constexpr int WO = POOL ? WI / 2 : WI;
            constexpr int OP = OUT_HALO ? lg_wp(WO) : WO;
            constexpr int OPL = OUT_HALO ? lg_plane(WO) : WO * WO;
            constexpr int OO = OUT_HALO ? lg_wp(WO) + 1 : 0;
            const int etau = 16 * tbw + (lane & 15);
            const int epb = etau / TPB, etl = etau % TPB;
            const int n = cur.n0 + epb;
            const int trg = cur.band * TROWS + etl / TC, tcg = etl % TC;
            // output transform of channel row r: 36 accumulator components -> the tile's 4 x 4 outputs (before ReLU)
            auto out_transform = [&](int r, float (&y)[4][4]) {
                float s[4][6];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const float m0 = acc[j][r], m1 = acc[6 + j][r], m2 = acc[12 + j][r], m3 = acc[18 + j][r], m4 = acc[24 + j][r],
                                m5 = acc[30 + j][r];
                    const float t1 = m1 + m2, t2 = m1 - m2, t3 = m3 + m4, t4 = m3 - m4;
                    s[0][j] = m0 + t1 + t3;
                    s[1][j] = fmaf(2.f, t4, t2);
                    s[2][j] = fmaf(4.f, t3, t1);
                    s[3][j] = fmaf(8.f, t4, t2) + m5;
                }
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float t1 = s[p][1] + s[p][2], t2 = s[p][1] - s[p][2], t3 = s[p][3] + s[p][4], t4 = s[p][3] - s[p][4];
                    y[p][0] = s[p][0] + t1 + t3;
                    y[p][1] = fmaf(2.f, t4, t2);
                    y[p][2] = fmaf(4.f, t3, t1);
                    y[p][3] = fmaf(8.f, t4, t2) + s[p][5];
                }
            };
            // ReLU (+ 2 x 2 max-pool) and the stores of channel row r
            auto out_store = [&](int r, const float (&y)[4][4]) {
                const int co = cur.cob * 64 + 16 * cb + 4 * (lane >> 4) + r;
                if ((LG_W4_EXP & 256) || ((LG_W4_EXP & 2048) && !IS_T) || ((LG_W4_EXP & 4096) && IS_T)) {
#pragma unroll
                    for (int p = 0; p < 4; p++) asm volatile("" ::"v"(y[p][0]), "v"(y[p][1]), "v"(y[p][2]), "v"(y[p][3]));
                } else if (n < N) {
                    float* o = out + ((size_t)n * COUT + co) * OPL + OO;
                    if (LG_W4_EXP & 512) o = out + (size_t)(co & 63) * OPL + OO;   // ablation: every item stores into the same 64 planes (L2 hits)
                    // One vector store per tile row (interior pixels start at an odd column: 4-byte aligned 8 / 16-byte stores,
                    // which the hardware takes): a wave instruction covers whole 64-128-byte runs of the plane.
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                    if (POOL) {
#pragma unroll
                        for (int p = 0; p < 2; p++) {
                            const float v0 = fmaxf(fmaxf(fmaxf(y[2 * p][0], y[2 * p][1]), fmaxf(y[2 * p + 1][0], y[2 * p + 1][1])), 0.f);
                            const float v1 = fmaxf(fmaxf(fmaxf(y[2 * p][2], y[2 * p][3]), fmaxf(y[2 * p + 1][2], y[2 * p + 1][3])), 0.f);
                            if (LG_W4_NT) __builtin_nontemporal_store((f32x2u){v0, v1}, reinterpret_cast<f32x2u*>(o + (2 * trg + p) * OP + 2 * tcg));
                            else *reinterpret_cast<f32x2u*>(o + (2 * trg + p) * OP + 2 * tcg) = (f32x2u){v0, v1};
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 4; p++)
                            if (LG_W4_NT) __builtin_nontemporal_store((f32x4u){fmaxf(y[p][0], 0.f), fmaxf(y[p][1], 0.f), fmaxf(y[p][2], 0.f), fmaxf(y[p][3], 0.f)},
                                                                      reinterpret_cast<f32x4u*>(o + (4 * trg + p) * OP + 4 * tcg));
                            else *reinterpret_cast<f32x4u*>(o + (4 * trg + p) * OP + 4 * tcg) =
                                (f32x4u){fmaxf(y[p][0], 0.f), fmaxf(y[p][1], 0.f), fmaxf(y[p][2], 0.f), fmaxf(y[p][3], 0.f)};
                    }
                    // The halo around the interior is (and stays) zero, and is written all the same: a cache line that keeps
                    // bytes its writer did not touch leaves the L2 as a masked write, which the memory side has to merge with
                    // the old line -- interior-only stores of layer 0's 1.6 GB took 0.81 ms, with the halo bytes 0.47 ms
                    // (tools/ubench/partial_lines.hip).  Row ends by the first / last tile of a tile row, the top and bottom
                    // rows (and corners) by the first / last tile row.  Only where it pays: layer 0 (32 x 32 output, 3 chunks per
                    // item: store bound) 0.77 -> 0.47 ms; the deeper layers hide their stores behind the matrix work and the extra
                    // single-lane store instructions cost them 0.01-0.10 ms each (16 x 16 and 8 x 8 maps are half edge tiles).
                    if (OUT_HALO && !POOL && WI >= 32 && !(LG_W4_EXP & 128)) {
                        typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
                        constexpr int TS = POOL ? 2 : 4;
                        const bool hl = tcg == 0, hr = tcg == TC - 1;
#pragma unroll
                        for (int p = 0; p < TS; p++) {
                            float* row = o + (TS * trg + p) * OP;
                            if (hl) row[-1] = 0.f;
                            if (hr) *reinterpret_cast<f32x3u*>(row + WO) = (f32x3u){0.f, 0.f, 0.f};
                        }
                        auto edge_row = [&](float* row) {
                            if (POOL) *reinterpret_cast<f32x2u*>(row + 2 * tcg) = (f32x2u){0.f, 0.f};
                            else *reinterpret_cast<f32x4u*>(row + 4 * tcg) = (f32x4u){0.f, 0.f, 0.f, 0.f};
                            if (hl) row[-1] = 0.f;
                            if (hr) *reinterpret_cast<f32x3u*>(row + WO) = (f32x3u){0.f, 0.f, 0.f};
                        };
                        if (trg == 0) edge_row(o - OP);
                        if (trg == TC - 1) edge_row(o + WO * OP);
                    }
                }
            };
            if (LG_W4_EXP & 64) {
#pragma unroll
                for (int p = 0; p < 36; p++) asm volatile("" ::"v"(acc[p]));
            } else if (SPLIT && NC >= 8 && P > 1 && it_i == R) {   // a split item (wave-uniform; both roles, all eight waves)
                // The output transform is linear: every part transforms ITS partial sums, and what travels is the 16 outputs
                // per channel and tile instead of the 36 accumulators (128 KB per part instead of 288), in registers the
                // accumulators have left free -- all 16 loads of a part are in flight at once.
                // The parts of an item run on one XCD and meet in ITS L2: no cache maintenance, only order.  (Agent-scope fences
                // -- __threadfence, release / acquire -- write the XCD's dirty lines back, megabytes of just-stored activations:
                // measured, they cost more than the split saves.)  A part's stores are acknowledged by the L2 (vmcnt 0) before
                // it counts itself in; the owner's compute unit has not read these addresses in this launch (its L1 was
                // invalidated when the kernel started), so its loads come from the L2.
                float yv[4][4][4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    out_transform(r, yv[r]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // [workgroup][channel row x tile row (16)][wave][lane] 16 bytes: a wave instruction moves 1 KB
                f32x4* mine = reinterpret_cast<f32x4*>(kpart) + (size_t)blockIdx.x * (16 * 512) + t;
                asm volatile("" : "+v"(mine));   // (opaque: hipcc otherwise forms the addresses in front of the item loop and spills them)
                if (cur.part != P - 1) {
#pragma unroll
                    for (int e = 0; e < 16; e++) mine[e * 512] = (f32x4){yv[e >> 2][e & 3][0], yv[e >> 2][e & 3][1], yv[e >> 2][e & 3][2], yv[e >> 2][e & 3][3]};
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    if (t == 0) __hip_atomic_fetch_add(&kflag[blockIdx.x + 8 * (P - 1 - cur.part)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;              // no stores, and no further item
                }
                if (t == 0) {
                    // (the parts have lower workgroup indices on this XCD: dispatched before this one, they are running or done.
                    //  Should that ever not hold, the wait ends after ~0.5 s, the item is stored from what has arrived and the word
                    //  at kerr -- host memory the library looks at after its next synchronisation -- makes that call fail.  No trap: a
                    //  faulting kernel can cost the whole node a reset.)
                    unsigned spins = 0;
                    while (__hip_atomic_load(&kflag[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(P - 1)) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) {
                            if (kerr) __hip_atomic_store(kerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                    __hip_atomic_store(&kflag[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next launch; every part has counted itself in)
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
                for (int k = 1; k < P; k++) {
                    const f32x4* theirs = mine - (size_t)(8 * k) * (16 * 512);   // workgroup jx - k of this XCD
                    asm volatile("" : "+v"(theirs));
                    f32x4 tmp[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) tmp[e] = __builtin_nontemporal_load(theirs + e * 512);
#pragma unroll
                    for (int e = 0; e < 16; e++)
#pragma unroll
                        for (int j = 0; j < 4; j++) yv[e >> 2][e & 3][j] += tmp[e][j];
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    out_store(r, yv[r]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float y[4][4];
                out_transform(r, y);
                out_store(r, y);
                __builtin_amdgcn_sched_barrier(0);   // one channel row at a time: keeps the transform temporaries of the 4 rows apart
            }
            prev_full = cur.n0 + PB <= N;            // every lane of this wave stored (n < N): the SCNT stores above were issued
#pragma unroll
            for (int p = 0; p < 36; p++) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (nxt.part == 0) acc[7] = *reinterpret_cast<const f32x4*>(&s_bias[cob_next * 64 + 16 * cb + 4 * (lane >> 4)]);
        }
    };
    if (wave < 4) role(std::true_type{});
    else role(std::false_type{});
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
    hipLaunchKernelGGL((lg_conv3x3_kernel<cfg.cin, cfg.cinp, cfg.cout, cfg.wi, cfg.pool, L != 5, KC, PP, CP>), dim3(grid),
                       dim3(256), 0, s, in, c->wconv[L], c->bconv[L], out, c->zeros);
}

// layer 0 of any encoder (9 input channels -> 64 or 128 padded output channels at 32x32): direct implicit GEMM
template <int COUT>
void launch_conv0(const float* in, const LgCnn* c, float* out, int N, hipStream_t s) {
    hipLaunchKernelGGL((lg_conv0_kernel<COUT>), dim3(N * (COUT / 64)), dim3(256), 0, s, in, c->wconv[0], c->bconv[0], out, c->zeros);
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

// out_halo: the next layer is a convolution (haloed plane); false: the head reads dense [C][npix]
bool launch_wino_rt(int cin, int cout, int wi, bool pool, bool out_halo, const float* in, const float* U, const float* bias,
                    float* out, int N, hipStream_t s) {
    const int tp = (wi / 2) * (wi / 2);
    const int ntb = tp >= 32 ? N * (tp / 32) : (N + 32 / tp - 1) / (32 / tp);
    const int grid = ntb * (cout / 64);
#define X(CI, CO, W_, P)                                                                                              \
    if (cin == CI && cout == CO && wi == W_ && pool == P) {                                                           \
        if (out_halo)                                                                                                 \
            hipLaunchKernelGGL((lg_wino_kernel<CI, CO, W_, P, true>), dim3(grid), dim3(256), 0, s, in, U, bias, out, N, ntb);  \
        else                                                                                                          \
            hipLaunchKernelGGL((lg_wino_kernel<CI, CO, W_, P, false>), dim3(grid), dim3(256), 0, s, in, U, bias, out, N, ntb); \
        return true;                                                                                                  \
    }
    LG_WINO_SHAPES(X)
#undef X
    return false;
}

#define LG_WINO4_L0_SHAPES(X) X(12, 64, 32, false) X(12, 128, 32, false)
// max_cus > 0: workgroups (= CUs) the persistent F(4x4) kernels may take (LgCnn::max_cus, the LG_CNN_CUS experiment)
// workgroups of the persistent F(4x4) kernels, at most (= CUs; a multiple of 8 so every XCD gets the same number)
static int lg_w4_num_cu() {
    static const int num_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n >= 8 ? n / 8 * 8 : 8;
    }();
    return num_cu;
}
constexpr size_t LG_W4_KPART_FLOATS = 16 * 512 * 4;   // partial outputs of one workgroup (lg_wino4_kernel, split items)

bool launch_wino4_rt(int cin, int cout, int wi, bool pool, bool out_halo, const float* in, const float* U4, const float* bias,
                     float* out, int N, int max_cus, float* kpart, unsigned* kflag, unsigned* kerr, hipStream_t s) {
    const int tp = (wi / 4) * (wi / 4);
    const int ntb = tp >= 32 ? N * (tp / 32) : (N + 32 / tp - 1) / (32 / tp);
    // persistent: one 512-thread workgroup per CU (140-156 KB of LDS each)
    const int num_cu = lg_w4_num_cu();
    const long long items8 = (long long)((ntb + 7) / 8) * (cout / 64);   // items per XCD
    const int cus = max_cus > 0 ? std::max(8, std::min(num_cu, max_cus / 8 * 8)) : num_cu;
    // Left-over items (fewer items than workgroups, or a last round that fills only some of them) are split along the input
    // channels over P workgroups each -- when that pays: the parts' exchange costs ~18 (P = 4) to ~26 us (P = 8), a part saves
    // (1 - 1/P) of an item's cin / 4 chunks of ~1.1 us (measured per layer at 20 and 640 patches, profiles/r04_cnn_split_items.txt:
    // 256 -> 256 channels halves at one frame and gains 18 % at 640 patches; 64-channel layers lose).
    int gx = cus / 8, P = 1;
    const int rem = items8 < gx ? (int)items8 : (int)(items8 % gx);
    if (rem > 0) {
        const int cap = std::min(std::min(gx / rem, 8), cin / 16);
        while (2 * P <= cap) P *= 2;
    }
    const bool split = P > 1 && (cin / 4) * (P - 1) / P >= 26;
    if (items8 < gx) gx = (int)items8 * (split ? P : 1);
    const int grid = 8 * gx;
#define X(CI, CO, W_, P_)                                                                                             \
    if (cin == CI && cout == CO && wi == W_ && pool == P_) {                                                          \
        constexpr bool CM = 36LL * CI * CO * 4 > 3 * 1024 * 1024;                                                     \
        constexpr bool CAN = CI >= 128;   /* (cin / 4 * 7 / 8 >= 26) */                                              \
        if (out_halo) {                                                                                               \
            if (CAN && split) hipLaunchKernelGGL((lg_wino4_kernel<CI, CO, W_, P_, true, CM, CAN>), dim3(grid), dim3(512), 0, s, in, U4, bias, out, N, ntb, kpart, kflag, kerr); \
            else hipLaunchKernelGGL((lg_wino4_kernel<CI, CO, W_, P_, true, CM, false>), dim3(grid), dim3(512), 0, s, in, U4, bias, out, N, ntb, kpart, kflag, kerr); \
        } else {                                                                                                      \
            if (CAN && split) hipLaunchKernelGGL((lg_wino4_kernel<CI, CO, W_, P_, false, CM, CAN>), dim3(grid), dim3(512), 0, s, in, U4, bias, out, N, ntb, kpart, kflag, kerr); \
            else hipLaunchKernelGGL((lg_wino4_kernel<CI, CO, W_, P_, false, CM, false>), dim3(grid), dim3(512), 0, s, in, U4, bias, out, N, ntb, kpart, kflag, kerr); \
        }                                                                                                             \
        return true;                                                                                                  \
    }
    LG_WINO_SHAPES(X)
    LG_WINO4_L0_SHAPES(X)
#undef X
    return false;
}

}  // namespace

bool lg_cnn_take_error(LgCnn* c) {
    if (!c->kerr_host || !*(volatile unsigned*)c->kerr_host) return false;
    *c->kerr_host = 0;
    return true;
}

void lg_cnn_free(LgCnn* c) {
    auto F = [](float*& p) { if (p) hipFree(p); p = nullptr; };
    for (int i = 0; i < 8; i++) { F(c->wconv[i]); F(c->bconv[i]); F(c->uwino[i]); F(c->uwino4[i]); F(c->act[i]); }
    F(c->att_w); F(c->ca_w1); F(c->ca_b1); F(c->ca_w2); F(c->ca_b2);
    for (int i = 0; i < 4; i++) { F(c->fcw[i]); F(c->fcb[i]); }
    F(c->in_halo); F(c->zeros); F(c->kpart);
    if (c->kflag) { hipFree(c->kflag); c->kflag = nullptr; }
    if (c->kerr_host) { hipHostFree(c->kerr_host); c->kerr_host = nullptr; c->kerr_dev = nullptr; }
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
            // Winograd-domain weights U = G g G^T of the BN-folded kernel (position 4i+j), rounded once from double, in the order
            // lg_wino_kernel's A fragments are read
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
                    // [k-step = ci/4][16-channel block][q = position group][lane = (ci%4)*16 + co%16][4 positions]
                    float* u = &uw[((((size_t)(ci / 4) * (l.coutp / 16) + co / 16) * 4) * 64 + (ci % 4) * 16 + co % 16) * 4];
                    for (int i = 0; i < 4; i++) {
                        u[256 * i + 0] = (float)gg[i][0];
                        u[256 * i + 1] = (float)(0.5 * (gg[i][0] + gg[i][1] + gg[i][2]));
                        u[256 * i + 2] = (float)(0.5 * (gg[i][0] - gg[i][1] + gg[i][2]));
                        u[256 * i + 3] = (float)gg[i][2];
                    }
                }
            }
            rc = upload(&c->uwino[L], uw, err);
            if (rc) return rc;
        }
        {
            // F(4x4,3x3) weights U = G g G^T (6x6) of the BN-folded kernel in the order lg_wino4_kernel's A fragments are read:
            // [k-step = ci/4][64-channel block][cb = (co%64)/16][position group pg = pos/4][lane = (ci%4)*16 + co%16][pos%4]
            static const double G4[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            const int cinp4 = L == 0 ? 12 : l.cinp;   // layer 0: 9 feature planes + 3 zero planes
            std::vector<float> u4((size_t)cinp4 * l.coutp * 36, 0.0f);
            const int ncb = l.coutp / 64;
            for (int co = 0; co < l.cout; co++) {
                for (int ci = 0; ci < l.cin; ci++) {
                    double g[3][3], gg[6][3];
                    for (int tap = 0; tap < 9; tap++)
                        g[tap / 3][tap % 3] = (double)w->conv_w[L][((size_t)co * l.cin + ci) * 9 + tap] * scv[co];
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 3; j++) gg[i][j] = G4[i][0] * g[0][j] + G4[i][1] * g[1][j] + G4[i][2] * g[2][j];
                    const size_t base = ((((size_t)(ci / 4) * ncb + co / 64) * 4 + (co % 64) / 16) * 9) * 256 + ((ci % 4) * 16 + co % 16) * 4;
                    for (int i = 0; i < 6; i++)
                        for (int j = 0; j < 6; j++) {
                            const int pos = 6 * i + j;
                            u4[base + (size_t)(pos / 4) * 256 + pos % 4] =
                                (float)(gg[i][0] * G4[j][0] + gg[i][1] * G4[j][1] + gg[i][2] * G4[j][2]);
                        }
                }
            }
            rc = upload(&c->uwino4[L], u4, err);
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
    {   // zeroed plane: padding channel of layer 0, unused staging slots
        std::vector<float> z(4096, 0.0f);
        int rc = upload(&c->zeros, z, err);
        if (rc) return rc;
    }
    // A/B and test switches of the standard encoder, read when the model is loaded (not on the per-call path):
    // LG_CNN_DIRECT=1 direct implicit GEMM for every layer; LG_CNN_WINO_MASK=<bits> bit L = layer L on Winograd
    // LG_CNN_F23=1 the F(2x2,3x3) Winograd kernels instead of F(4x4,3x3)
    c->use_f23 = getenv("LG_CNN_F23") != nullptr;
    c->wino_mask = getenv("LG_CNN_DIRECT") ? 0 : 0x3f;
    if (const char* e = getenv("LG_CNN_WINO_MASK")) c->wino_mask = atoi(e) & 0x3f;
    c->loaded = true;
    return LG_OK;
}


// haloed input patch: the 9 feature planes + 3 zero planes (layer 0 runs on the Winograd kernel in chunks of 4 channels)
size_t lg_cnn_halo_patch_floats(void) { return (size_t)12 * lg_plane(32); }

// activation buffers, one per layer (fixed geometry: the zero halos are written once, at allocation)
static int ensure_act(LgCnn* c, int N, hipStream_t s, std::string* err) {
    if (!c->kpart) {   // partial sums + counters of the split items of lg_wino4_kernel (one slot per workgroup; the kernel resets its counters)
        const size_t wg = (size_t)lg_w4_num_cu();
        if (hipMalloc((void**)&c->kpart, wg * LG_W4_KPART_FLOATS * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&c->kflag, wg * sizeof(unsigned)) != hipSuccess ||
            hipMemsetAsync(c->kflag, 0, wg * sizeof(unsigned), s) != hipSuccess) {
            *err = "lg_cnn_forward: workspace allocation failed";
            return LG_ERR_NOMEM;
        }
        // the word a split item sets when its parts do not arrive: pinned host memory the kernels write directly (no mapping: no word)
        if (hipHostMalloc((void**)&c->kerr_host, sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
            *c->kerr_host = 0;
            if (hipHostGetDevicePointer((void**)&c->kerr_dev, c->kerr_host, 0) != hipSuccess) c->kerr_dev = nullptr;
        } else {
            (void)hipGetLastError();
            c->kerr_host = nullptr;
        }
    }
    if (N <= c->capN) return LG_OK;
    hipStreamSynchronize(s);
    for (int i = 0; i < 8; i++)
        if (c->act[i]) { hipFree(c->act[i]); c->act[i] = nullptr; }
    if (c->in_halo) { hipFree(c->in_halo); c->in_halo = nullptr; }
    c->capN = 0;
    bool ok = true;
    for (int L = 0; L < c->n_layers && ok; L++) {
        const RtLayer& l = c->layers[L];
        const int wo = l.pool ? l.wi / 2 : l.wi;
        const size_t per = (size_t)l.coutp * (L + 1 < c->n_layers ? lg_plane(wo) : wo * wo);
        c->act_per[L] = per;
        ok = hipMalloc((void**)&c->act[L], (size_t)N * per * sizeof(float)) == hipSuccess &&
             hipMemsetAsync(c->act[L], 0, (size_t)N * per * sizeof(float), s) == hipSuccess;
    }
    ok = ok && hipMalloc((void**)&c->in_halo, (size_t)N * lg_cnn_halo_patch_floats() * sizeof(float)) == hipSuccess &&
         hipMemsetAsync(c->in_halo, 0, (size_t)N * lg_cnn_halo_patch_floats() * sizeof(float), s) == hipSuccess;
    if (!ok) {
        *err = "lg_cnn_forward: activation workspace allocation failed";
        return LG_ERR_NOMEM;
    }
    c->capN = N;
    return LG_OK;
}

static int lg_cnn_run_slice(LgCnn* c, const float* patches, bool haloed_in, int N, float* logits, hipStream_t s, std::string* err);

// Slices bound the activation workspace (0.8 MB per patch for the standard model): at most 8192 patches at a time.
int lg_cnn_run(LgCnn* c, const float* patches, bool haloed_in, int N, float* logits, hipStream_t s, std::string* err) {
    if (!c->loaded) { *err = "no model"; return LG_ERR_NO_MODEL; }
    const int max_slice = 8192;
    const size_t pstride = haloed_in ? lg_cnn_halo_patch_floats() : (size_t)9 * 1024;
    for (int off = 0; off < N; off += max_slice) {
        const int n = N - off < max_slice ? N - off : max_slice;
        int rc = lg_cnn_run_slice(c, patches + (size_t)off * pstride, haloed_in, n, logits + off, s, err);
        if (rc) return rc;
    }
    return LG_OK;
}

static int lg_cnn_run_slice(LgCnn* c, const float* patches, bool haloed_in, int N, float* logits, hipStream_t s, std::string* err) {
    int rc = ensure_act(c, N, s, err);
    if (rc) return rc;
    const float* x = patches;
    if (!haloed_in) {   // the C-ABI's dense [N][9][32][32] patches -> haloed planes
        hipLaunchKernelGGL(lg_repack_kernel, dim3((unsigned)(N * 9)), dim3(256), 0, s, patches, c->in_halo, (long long)N * 9);
        x = c->in_halo;
    }
    // layer 0: the F(4x4,3x3) kernel on 12 input planes (3 chunks); the direct 9-channel kernel with LG_CNN_DIRECT / _F23 / mask bit 0 clear
    if (!c->use_f23 && (c->wino_mask & 1) &&
        launch_wino4_rt(12, c->layers[0].coutp, 32, false, true, x, c->uwino4[0], c->bconv[0], c->act[0], N, c->max_cus, c->kpart, c->kflag, c->kerr_dev, s)) {
        // (a shape the F(4x4) table lacks falls through to the direct kernel instead of leaving act[0] unwritten)
    } else if (c->layers[0].coutp == 64) launch_conv0<64>(x, c, c->act[0], N, s);
    else launch_conv0<128>(x, c, c->act[0], N, s);
    const float* cur = c->act[0];
    if (c->standard) {
        const int wmask = c->wino_mask;
#define LG_LAYER(L, KC, PP, CP)                                                                                     \
    do {                                                                                                            \
        const RtLayer& l = c->layers[L];                                                                            \
        if (!(wmask & (1 << L))) launch_conv<L, KC, PP, CP>(cur, c, c->act[L], N, s);                               \
        else if (c->use_f23) launch_wino_rt(l.cinp, l.coutp, l.wi, l.pool, L != 5, cur, c->uwino[L], c->bconv[L], c->act[L], N, s); \
        else launch_wino4_rt(l.cinp, l.coutp, l.wi, l.pool, L != 5, cur, c->uwino4[L], c->bconv[L], c->act[L], N, c->max_cus, c->kpart, c->kflag, c->kerr_dev, s); \
        cur = c->act[L];                                                                                            \
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
            const bool okl = c->use_f23 ? launch_wino_rt(l.cinp, l.coutp, l.wi, l.pool, L + 1 < c->n_layers, cur, c->uwino[L], c->bconv[L], c->act[L], N, s)
                                        : launch_wino4_rt(l.cinp, l.coutp, l.wi, l.pool, L + 1 < c->n_layers, cur, c->uwino4[L], c->bconv[L], c->act[L], N, c->max_cus, c->kpart, c->kflag, c->kerr_dev, s);
            if (!okl) {
                *err = "lg_cnn_forward: unsupported layer shape";
                return LG_ERR_UNSUPPORTED;
            }
            cur = c->act[L];
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
