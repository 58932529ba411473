// tools/micro/pk_fma_bench.hip -- does packed FP32 (v_pk_fma_f32) raise the FMA rate of a gfx950 SIMD?
// Each lane runs CH independent dependent-FMA chains; the scalar build issues CH v_fma_f32 per iteration, the packed build
// CH/2 v_pk_fma_f32 (the same FLOPs).  W waves per SIMD on every SIMD of the chip.  Prints GFLOP/s and cycles per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/pk_fma_bench.hip -o /tmp/pk_fma_bench && /tmp/pk_fma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int CH = 16, ITERS = 8192;

__global__ void __launch_bounds__(64) k_scalar(float* out, float x, float y) {
    float a[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) a[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(64) k_packed(float* out, float x, float y) {
    v2f a[CH / 2];
    const v2f xx = {x, x}, yy = {y, y};
#pragma unroll
    for (int i = 0; i < CH / 2; i++) a[i] = v2f{(float)(threadIdx.x + 2 * i), (float)(threadIdx.x + 2 * i + 1)};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH / 2; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(xx), "v"(yy));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH / 2; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
// the mix the render kernel's step body has: a packed pair needs its operands in an aligned register pair; when the two halves
// come from separate scalar computations the compiler adds moves.  2 v_mov_b32 + 1 v_pk_fma against 2 v_fma.
__global__ void __launch_bounds__(64) k_packed_mov(float* out, float x, float y) {
    v2f a[CH / 2];
    const v2f xx = {x, x}, yy = {y, y};
#pragma unroll
    for (int i = 0; i < CH / 2; i++) a[i] = v2f{(float)(threadIdx.x + 2 * i), (float)(threadIdx.x + 2 * i + 1)};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH / 2; i++) {
            float lo = a[i].x, hi = a[i].y;
            asm volatile("v_mov_b32 %0, %0" : "+v"(lo));
            asm volatile("v_mov_b32 %0, %0" : "+v"(hi));
            a[i] = v2f{lo, hi};
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(xx), "v"(yy));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH / 2; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;   // Hz
    printf("%s: %d CUs, %.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
    float* out; hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w : {1, 2, 4, 8}) {
        const int grid = cus * 4 * w;
        for (int v = 0; v < 3; v++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(64), 0, 0, out, 0.999f, 0.001f);
                else if (v == 1) hipLaunchKernelGGL(k_packed, dim3(grid), dim3(64), 0, 0, out, 0.999f, 0.001f);
                else hipLaunchKernelGGL(k_packed_mov, dim3(grid), dim3(64), 0, 0, out, 0.999f, 0.001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double fma_per_lane = (double)ITERS * CH;                       // scalar-FMA equivalents per lane
            const double flops = 2.0 * fma_per_lane * 64.0 * grid;
            const double inst_per_wave = v == 0 ? fma_per_lane : (v == 1 ? fma_per_lane / 2 : fma_per_lane * 1.5);
            // cycles one SIMD spends per wave-instruction it issues: time * clock / (instructions per wave * waves on the SIMD)
            const double cyc = (double)best * 1e-3 * clk / (inst_per_wave * w);
            printf("%d waves/SIMD  %-26s %8.3f ms  %9.1f GFLOP/s  %.2f SIMD-cycles per wave-instruction (at the nominal clock)\n", w,
                   v == 0 ? "v_fma_f32" : v == 1 ? "v_pk_fma_f32" : "2 v_mov + v_pk_fma_f32", best, flops / (best * 1e-3) / 1e9, cyc);
        }
    }
    return 0;
}
