// Micro-benchmark: what HBM rate does the score-plane kernel's TRAFFIC MIX reach on gfx950, without any of its arithmetic?
// Per pixel: read depth (4 B) and distance (4 B), write seven float planes (28 B) and one byte plane (1 B) = 37 B -- the mix of
// lg_final_kernel's dense path (37.25 B with the bit rows).  Variants:
//   linear    each thread 4 consecutive pixels of the flattened batch, grid-stride (a wave = 1 KiB contiguous per plane)
//   tiled     the kernel's shape: 64 x 16 pixel tiles, 256 threads, a wave = 4 rows x 256 B per plane, XCD-contiguous tile order
//   tiled128  128 x 8 pixel tiles (a wave = 2 rows x 512 B)
//   no byte plane: tiled 64 x 16 without the 1-byte plane (its 64-byte runs are half cache lines)
//   writes    tiled, the nine stores only (the constant-tile path of the kernel)
//   copy      linear, 2 reads + 1 write of 4 B (a plain streaming reference)
// Prints ms and TB/s per variant (median of `reps` launches).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/stream_mix tools/ubench/stream_mix.hip
// Run:   tools/ubench/stream_mix [frames=128] [reps=7]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

struct Planes { float* p[7]; uint8_t* valid; const float* depth; const float* dist; };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool VALID = true>
__device__ __forceinline__ void emit(const Planes& P, size_t off, f4 a, f4 b, bool reads) {
    f4 v = reads ? a + b : (f4){1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 7; i++) *reinterpret_cast<f4*>(P.p[i] + off) = v + (float)i;
    if (VALID) __builtin_nontemporal_store((uint32_t)(v.x > 0.5f ? 0x01010101u : 0u), reinterpret_cast<uint32_t*>(P.valid + off));
}

__global__ __launch_bounds__(256) void k_linear(Planes P, size_t npx) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < npx; i += (size_t)gridDim.x * 1024) {
        const f4 a = *reinterpret_cast<const f4*>(P.depth + i), b = *reinterpret_cast<const f4*>(P.dist + i);
        emit(P, i, a, b, true);
    }
}

__global__ __launch_bounds__(256) void k_copy(Planes P, size_t npx) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < npx; i += (size_t)gridDim.x * 1024) {
        const f4 a = *reinterpret_cast<const f4*>(P.depth + i), b = *reinterpret_cast<const f4*>(P.dist + i);
        *reinterpret_cast<f4*>(P.p[0] + i) = a + b;
    }
}

// TW x TH tiles, 256 threads x 4 pixels; resident workgroups walk an XCD-contiguous range (blockIdx % 8 = XCD)
template <int TW, int TH, bool READS, bool VALID = true>
__global__ __launch_bounds__(256) void k_tiled(Planes P, int B, int H, int W) {
    static_assert(TW * TH == 1024, "256 threads x 4 pixels");
    const int tiles_x = W / TW, tiles_y = (H + TH - 1) / TH, ntile = tiles_x * tiles_y;
    const long long total = (long long)ntile * B, q = total / 8, r = total % 8;
    const long long xcd = blockIdx.x % 8;
    const long long first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, count = q + (xcd < r ? 1 : 0);
    const int lx = (threadIdx.x % (TW / 4)) * 4, ly = threadIdx.x / (TW / 4);
    for (long long it = blockIdx.x / 8; it < count; it += (gridDim.x + 7) / 8) {
        const long long id = first + it;
        const int frame = (int)(id / ntile), tile = (int)(id % ntile), bx = tile % tiles_x, by = tile / tiles_x;
        const int y = by * TH + ly, x = bx * TW + lx;
        if (y >= H) continue;
        const size_t off = ((size_t)frame * H + y) * W + x;
        f4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (READS) { a = *reinterpret_cast<const f4*>(P.depth + off); b = *reinterpret_cast<const f4*>(P.dist + off); }
        emit<VALID>(P, off, a, b, READS);
    }
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; i++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, reps = argc > 2 ? atoi(argv[2]) : 7;
    const int H = 1080, W = 1920;
    const size_t npx = (size_t)B * H * W;
    Planes P;
    for (int i = 0; i < 7; i++) CHECK(hipMalloc((void**)&P.p[i], npx * 4));
    CHECK(hipMalloc((void**)&P.valid, npx));
    float *depth, *dist;
    CHECK(hipMalloc((void**)&depth, npx * 4)); CHECK(hipMalloc((void**)&dist, npx * 4));
    CHECK(hipMemset(depth, 0, npx * 4)); CHECK(hipMemset(dist, 0, npx * 4));
    P.depth = depth; P.dist = dist;
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = cus * 8;
    struct { const char* name; double bytes_per_px; double ms; } res[] = {
        {"linear   2R + 7W4 + 1W1", 37.0, time_ms([&] { hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, P, npx); }, reps)},
        {"tiled    64x16  same mix", 37.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<64, 16, true>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"tiled    128x8  same mix", 37.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<128, 8, true>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"tiled    256x4  same mix", 37.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<256, 4, true>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"tiled    64x16  no byte plane", 36.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<64, 16, true, false>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"writes   64x16  7W4 + 1W1", 29.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<64, 16, false>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"writes   256x4  7W4 + 1W1", 29.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<256, 4, false>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"writes   128x8  7W4 + 1W1", 29.0, time_ms([&] { hipLaunchKernelGGL((k_tiled<128, 8, false>), dim3(grid), dim3(256), 0, 0, P, B, H, W); }, reps)},
        {"copy     2R + 1W4", 12.0, time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, P, npx); }, reps)},
    };
    printf("frames=%d (%d x %d), %d resident workgroups of 256 threads, median of %d launches\n", B, H, W, grid, reps);
    for (auto& r : res) printf("%-28s %8.3f ms  %6.2f TB/s  (%.3f of 8 TB/s)\n", r.name, r.ms, r.bytes_per_px * npx / (r.ms * 1e-3) / 1e12,
                               r.bytes_per_px * npx / (r.ms * 1e-3) / 8e12);
    return 0;
}
