// Micro-benchmark: which lane -> address patterns of ds_read_b128 / ds_read_b64 run without bank conflicts on gfx950?
// One kernel per pattern (template), REPS reads per wave; run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- tools/ubench/lds_conflict
// and divide the counters by the instruction count (tools/pmc_lds_ubench.sh).  Patterns: the tile loads of lg_wino4_kernel's
// input transform (lane = (channel bit, tile)) for image sizes 32 / 16 / 8 with the LDS row pitch WP, the patch stride RS and
// the channel stride S (floats) as parameters; KIND 0: ds_read_b128 at the tile's first column, 1: ds_read_b128 four columns
// further (the tile's columns 4..7), 2: ds_read_b64 there (what the kernel did for columns 4..5).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/lds_conflict tools/ubench/lds_conflict.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int REPS = 4096;

template <int WI, int WP, int RS, int S>
__device__ int tile_off(int l) {
    constexpr int TC = WI / 4, TP = TC * TC, PB = TP >= 32 ? 1 : 32 / TP, TPB = 32 / PB;
    const int tk = l >> 5, tau = l & 31;
    return tk * S + (tau / TPB) * RS + (4 * ((tau % TPB) / TC)) * WP + 4 * ((tau % TPB) % TC);
}

// PERM: which 8-lane group (bits 3..4 of the lane) holds which group of tiles, as a base-4 number g0 g1 g2 g3 (0123 = identity)
template <int WI, int WP, int RS, int S, int KIND, int PERM = 123>
__global__ __launch_bounds__(256) void k(float* out) {
    __shared__ __attribute__((aligned(16))) float s[16384];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 16384; i += 256) s[i] = (float)i;
    __syncthreads();
    const int pg[4] = {PERM / 1000, (PERM / 100) % 10, (PERM / 10) % 10, PERM % 10};
    const int pl = (lane & ~24) | (pg[(lane >> 3) & 3] << 3);
    const int off = (WI ? tile_off<WI ? WI : 8, WP, RS, S>(pl) : 4 * pl) + (KIND ? 4 : 0);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int r = 0; r < REPS; r++) {
        const int o = (off + WP * (r & 3)) & 16380;   // rows of the tile: the pattern between lanes is unchanged
        if (KIND < 2) {
            f4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(o * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += v;
        } else {
            f2 v;
            asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(o * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc.x += v.x; acc.y += v.y;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) out[t] = acc.x;
}

// KIND 3: the transform's whole load sequence as the kernel issues it -- four waves (channel pair, half), ten 16-byte reads
// back to back (rows 0..4 of the tile at columns 0..3 and 4..7), one wait
template <int WI, int WP, int RS, int S>
__global__ __launch_bounds__(256) void kseq(float* out) {
    __shared__ __attribute__((aligned(16))) float s[16384];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 16384; i += 256) s[i] = (float)i;
    __syncthreads();
    const int tk2 = (wave >> 1) << 1, th = wave & 1;
    const int off = tile_off<WI, WP, RS, S>(lane) + tk2 * S + th * WP;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < REPS / 8; r++) {
        const int o = ((off + 4096 * (r & 1)) & 16380) * 4;
        f4 v[10];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[2 * q]) : "v"(o), "i"(q * WP * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[2 * q + 1]) : "v"(o), "i"(q * WP * 4 + 16) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 10; q++) acc += v[q];
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) out[t] = acc.x;
}

template <int WI, int WP, int RS, int S, int PERM>
void runp(float* d) {
    hipLaunchKernelGGL((k<WI, WP, RS, S, 0, PERM>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<WI, WP, RS, S, 1, PERM>), dim3(128), dim3(256), 0, 0, d);
}
template <int WI, int WP, int RS, int S>
void run(float* d) {
    hipLaunchKernelGGL((k<WI, WP, RS, S, 0>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<WI, WP, RS, S, 1>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<WI, WP, RS, S, 2>), dim3(128), dim3(256), 0, 0, d);
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, 1 << 20));
    run<0, 4, 0, 0>(d);                                   // contiguous 16-byte pieces
    run<32, 36, 648, 648>(d);                             // the kernel's layouts until round 3
    run<16, 20, 360, 720>(d);
    run<8, 12, 120, 960>(d);
    run<32, 40, 720, 720>(d); run<32, 40, 720, 736>(d); run<32, 44, 792, 792>(d);
    run<16, 24, 432, 864>(d); run<16, 24, 432, 880>(d); run<16, 20, 368, 736>(d); run<16, 20, 360, 736>(d); run<16, 20, 376, 752>(d);
    run<8, 12, 120, 976>(d); run<8, 12, 120, 992>(d); run<8, 12, 124, 992>(d); run<8, 12, 124, 1008>(d); run<8, 12, 128, 1024>(d);
    run<8, 12, 128, 1040>(d); run<8, 12, 132, 1056>(d); run<8, 12, 136, 1088>(d); run<8, 12, 136, 1104>(d); run<8, 12, 120, 1008>(d);
    run<8, 12, 124, 1024>(d); run<8, 12, 132, 1072>(d); run<8, 12, 140, 1120>(d); run<8, 12, 144, 1152>(d); run<8, 12, 144, 1168>(d);
    // the kernel's own pitches with the 8-lane groups of the transform dealt differently to the tiles: all 24 orders
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 132>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 213>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 231>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 312>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 321>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1023>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1032>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1203>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1230>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1302>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 1320>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2013>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2031>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2103>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2130>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2301>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 2310>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3012>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3021>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3102>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3120>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3201>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<32, 36, 648, 648, 0, 3210>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 132>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 213>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 231>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 312>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 321>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1023>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1032>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1203>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1230>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1302>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 1320>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2013>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2031>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2103>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2130>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2301>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 2310>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3012>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3021>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3102>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3120>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3201>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<16, 20, 360, 720, 0, 3210>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 132>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 213>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 231>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 312>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 321>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1023>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1032>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1203>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1230>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1302>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 1320>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2013>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2031>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2103>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2130>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2301>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 2310>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3012>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3021>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3102>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3120>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3201>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((k<8, 12, 120, 960, 0, 3210>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((kseq<32, 36, 648, 648>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((kseq<32, 40, 720, 720>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((kseq<16, 20, 360, 720>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((kseq<16, 24, 432, 864>), dim3(128), dim3(256), 0, 0, d);
    hipLaunchKernelGGL((kseq<8, 12, 120, 960>), dim3(128), dim3(256), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    printf("ok: 128 workgroups x 4 waves x %d instructions per kernel\n", REPS);
    return 0;
}
