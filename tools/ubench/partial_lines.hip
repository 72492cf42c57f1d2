// Micro-benchmark: what do stores that leave part of every cache line unwritten cost on gfx950?
// The CNN's activation planes are haloed: a 32 x 32 map lives in 34 rows x 36 floats (144 B per row, the plane contiguous), the
// halo is zeroed once and the layer kernels only store the 32 x 32 interior -- 128-byte runs that start 4 bytes into a row, so
// every 128-byte line of the plane keeps some bytes the kernel never writes.  Variants over `planes` planes (default: the 1.6 GB
// of layer 0's output for 5120 patches):
//   interior   16 B per lane at byte 4 + 16 s of rows 1..32 (what the layer kernels do)
//   +halo      the same stores plus the row's 4 + 12 halo bytes (zeros) from the first / last lane of the row, and rows 0 / 33
//   full       the whole plane as 306 contiguous 16-byte pieces (every line written completely, aligned)
// Prints ms and the rate of USEFUL bytes (the interior: 4096 B per plane) and of touched bytes.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/partial_lines tools/ubench/partial_lines.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 34, PITCH = 36, PLANE = ROWS * PITCH;   // floats

// one workgroup of 256 threads per plane: thread = (row 1..32, segment 0..7)
template <bool HALO>
__global__ __launch_bounds__(256) void k_interior(float* buf, int planes) {
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float* pl = buf + (size_t)p * PLANE;
        const int row = 1 + (threadIdx.x >> 3), seg = threadIdx.x & 7;
        float* dst = pl + row * PITCH + 1 + 4 * seg;
        const float v = (float)(p + threadIdx.x);
        *reinterpret_cast<f4*>(dst) = (f4){v, v, v, v};
        if (HALO) {
            if (seg == 0) pl[row * PITCH] = 0.0f;
            if (seg == 7) *reinterpret_cast<f3*>(pl + row * PITCH + 33) = (f3){0.f, 0.f, 0.f};
            if (threadIdx.x < 18) {   // rows 0 and 33: 2 x 9 pieces of 16 bytes
                const int r = threadIdx.x < 9 ? 0 : 33, c = threadIdx.x % 9;
                *reinterpret_cast<f4*>(pl + r * PITCH + 4 * c) = (f4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_full(float* buf, int planes) {
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float* pl = buf + (size_t)p * PLANE;
        const float v = (float)(p + threadIdx.x);
        for (int c = threadIdx.x; c < PLANE / 4; c += 256) *reinterpret_cast<f4*>(pl + 4 * c) = (f4){v, v, v, v};
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
    const int planes = argc > 1 ? atoi(argv[1]) : 5120 * 64, reps = argc > 2 ? atoi(argv[2]) : 7;
    float* buf;
    CHECK(hipMalloc((void**)&buf, (size_t)planes * PLANE * 4));
    CHECK(hipMemset(buf, 0, (size_t)planes * PLANE * 4));
    const int grid = 256 * 8;
    const double useful = (double)planes * 32 * 32 * 4, all = (double)planes * PLANE * 4;
    struct { const char* name; double touched, ms; } res[] = {
        {"interior", useful, time_ms([&] { hipLaunchKernelGGL(k_interior<false>, dim3(grid), dim3(256), 0, 0, buf, planes); }, reps)},
        {"+halo", all, time_ms([&] { hipLaunchKernelGGL(k_interior<true>, dim3(grid), dim3(256), 0, 0, buf, planes); }, reps)},
        {"full", all, time_ms([&] { hipLaunchKernelGGL(k_full, dim3(grid), dim3(256), 0, 0, buf, planes); }, reps)},
    };
    printf("%d planes of %d x %d floats (%.2f GB, interior %.2f GB), median of %d launches\n", planes, ROWS, PITCH, all / 1e9, useful / 1e9, reps);
    for (auto& r : res)
        printf("%-10s %8.3f ms   useful %5.2f TB/s   touched %5.2f TB/s\n", r.name, r.ms, useful / (r.ms * 1e-3) / 1e12, r.touched / (r.ms * 1e-3) / 1e12);
    return 0;
}
