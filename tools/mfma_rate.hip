// fp32 MFMA issue-rate microbenchmark for gfx950 (used to calibrate the rooflines in DESIGN.md / profiles):
//   hipcc -O3 --offload-arch=gfx950 -o mfma_rate tools/mfma_rate.hip && ./mfma_rate
// Measured on MI355X: v_mfma_f32_16x16x4_f32 32.6 cycles back to back (independent accumulators),
// v_mfma_f32_32x32x2_f32 64.4; full chip 147.6 / 149.2 TFLOP/s (spec 157.3); v_fma_f32 3.8 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + a;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = threadIdx.x * 0.001f, b = 1.0f + a;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
__global__ void kfma(float* out, unsigned long long* cyc, int iters) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    float a = threadIdx.x * 0.001f, b = 1.0f + a;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(a, b, acc[i]);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[2] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4 * sizeof(float)); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    const int iters = 1000;
    for (int waves = 1; waves <= 2; ++waves) {
        for (int grid : {1, 256, 1024}) {
            hipLaunchKernelGGL(k16, dim3(grid), dim3(256 * waves), 0, 0, out, cyc, iters);
            hipLaunchKernelGGL(k32, dim3(grid), dim3(256 * waves), 0, 0, out, cyc, iters);
            hipLaunchKernelGGL(kfma, dim3(grid), dim3(256 * waves), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            unsigned long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
            printf("waves/SIMD %d grid %4d: 16x16x4: %.1f cyc/MFMA (%.1f flop/cyc/SIMD)   32x32x2: %.1f cyc/MFMA (%.1f flop/cyc/SIMD)   v_fma: %.2f cyc/instr\n",
                   waves, grid, h[0] / (16.0 * iters) / 1, 2048.0 * waves / (h[0] / (16.0 * iters)), h[1] / (4.0 * iters), 4096.0 * waves / (h[1] / (4.0 * iters)), h[2] / (16.0 * iters));
        }
    }
    // wall-clock TFLOP/s at full chip
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k16, dim3(1024), dim3(256), 0, 0, out, cyc, 4000);
        else hipLaunchKernelGGL(k32, dim3(1024), dim3(256), 0, 0, out, cyc, 4000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 1024.0 * 4 * 4000 * (which == 0 ? 16 * 2048.0 : 4 * 4096.0);
        printf("%s full chip: %.3f ms -> %.1f TFLOP/s\n", which == 0 ? "16x16x4" : "32x32x2", ms, flops / ms / 1e9);
    }
    return 0;
}
