// Micro-benchmark: what does one extra instruction cost beside back-to-back v_mfma_f32_16x16x4_f32 on gfx950?
// Each wave runs ITERS x (16 independent MFMAs, with F filler instructions of one kind after each MFMA) and stamps
// s_memtime around the loop; the host prints the median cycles per MFMA slot for 1 and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/mfma_issue tools/ubench/mfma_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { K_NONE, K_VADD, K_VFMA, K_PKADD, K_DSR128, K_DSR64, K_DSW128, K_GLD128, K_DMA4, K_DMA16, K_NKINDS };
static const char* kNames[] = {"none", "v_add_f32", "v_fma_f32", "v_pk_add_f32", "ds_read_b128", "ds_read_b64", "ds_write_b128",
                               "global_load_dwordx4(L2)", "global_load_lds_dword", "global_load_lds_dwordx4"};

template <int KIND, int F, int MF>   // MF = 1: MFMAs present, 0: fillers only
__global__ void k_issue(const float* __restrict__ gsrc, long long* __restrict__ cyc, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lane * 0.25f + i;
    f32x2 pk[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pk[i] = (f32x2){(float)lane, (float)i};
    f32x4 ld[4];
#pragma unroll
    for (int i = 0; i < 4; i++) ld[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const float* lp = lds + (t * 4) % 4096;
    const float* gp = gsrc + ((size_t)blockIdx.x * 64 + lane) * 4 % 65536;
    float* dma_dst = lds + 4096 + wave * 256;   // 1 KiB per wave

    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (MF) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; f++) {
                const int s = (g * F + f);
                if (KIND == K_VADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[s & 7]) : "v"(a));
                if (KIND == K_VFMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[s & 7]) : "v"(a), "v"(b));
                if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[s & 3]) : "v"(pk[(s + 1) & 3]));
                if (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[s & 3]) : "v"((unsigned)(size_t)(lp)) : "memory");
                if (KIND == K_DSR64) asm volatile("ds_read_b64 %0, %1" : "=v"(pk[s & 3]) : "v"((unsigned)(size_t)(lp)) : "memory");
                if (KIND == K_DSW128) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(lp)), "v"(ld[s & 3]) : "memory");
                if (KIND == K_GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[s & 3]) : "v"(gp) : "memory");
                if (KIND == K_DMA4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp),
                                                     (__attribute__((address_space(3))) void*)(dma_dst), 4, 0, 0);
                if (KIND == K_DMA16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp),
                                                     (__attribute__((address_space(3))) void*)(dma_dst), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += pk[i][0] + pk[i][1] + ld[i][0] + ld[i][3];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int KIND, int F, int MF>
void run(int threads, const float* gsrc, long long* d_cyc, float* sink, int iters) {
    const int grid = 256;
    std::vector<long long> h(grid * threads / 64);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_issue<KIND, F, MF>), dim3(grid), dim3(threads), 8192 * 4, 0, gsrc, d_cyc, sink, iters);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_issue<KIND, F, MF>), dim3(grid), dim3(threads), 8192 * 4, 0, gsrc, d_cyc, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // per SIMD: waves_per_simd waves share it; cycles per MFMA slot per SIMD = med / (iters*16) / waves_per_simd
    const int wps = threads / 256;
    printf("%-26s F=%d mfma=%d waves/SIMD=%d : %7.2f cyc per slot per wave, %7.2f per slot per SIMD, wall %.3f ms\n", kNames[KIND], F, MF,
           wps, med / (iters * 16.0), med / (iters * 16.0) / wps, ms);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int KIND>
void sweep(const float* gsrc, long long* d_cyc, float* sink, int iters) {
    for (int threads : {256, 512}) {
        run<KIND, 1, 1>(threads, gsrc, d_cyc, sink, iters);
        run<KIND, 2, 1>(threads, gsrc, d_cyc, sink, iters);
        run<KIND, 4, 1>(threads, gsrc, d_cyc, sink, iters);
        run<KIND, 4, 0>(threads, gsrc, d_cyc, sink, iters);
    }
}

int main() {
    float *gsrc, *sink;
    long long* d_cyc;
    hipMalloc((void**)&gsrc, 65536 * 4 + 4096);
    hipMemset(gsrc, 0, 65536 * 4 + 4096);
    hipMalloc((void**)&sink, 16);
    hipMalloc((void**)&d_cyc, 256 * 8 * sizeof(long long));
    const int iters = 2000;
    run<K_NONE, 0, 1>(256, gsrc, d_cyc, sink, iters);
    run<K_NONE, 0, 1>(512, gsrc, d_cyc, sink, iters);
    sweep<K_VADD>(gsrc, d_cyc, sink, iters);
    sweep<K_VFMA>(gsrc, d_cyc, sink, iters);
    sweep<K_PKADD>(gsrc, d_cyc, sink, iters);
    sweep<K_DSR128>(gsrc, d_cyc, sink, iters);
    sweep<K_DSR64>(gsrc, d_cyc, sink, iters);
    sweep<K_DSW128>(gsrc, d_cyc, sink, iters);
    sweep<K_GLD128>(gsrc, d_cyc, sink, iters);
    sweep<K_DMA4>(gsrc, d_cyc, sink, iters);
    sweep<K_DMA16>(gsrc, d_cyc, sink, iters);
    return 0;
}
