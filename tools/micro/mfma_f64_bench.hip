// tools/micro/mfma_f64_bench.hip — how fast are v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, alone and side by side?
// Independent accumulators, one to eight waves per CU; prints cycles per instruction per wave (s_memtime) and the chip-wide TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_bench tools/micro/mfma_f64_bench.hip && ./mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double acc_t __attribute__((ext_vector_type(4)));
// (launch bound 512 = a 256-register budget: with 512 registers the compiler parks the accumulators in AGPRs and copies all of
// them to and from VGPRs around every MFMA of the loop - that variant measured 141 cycles per MFMA, the copies, not the pipe)
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma(double* out, unsigned long long* cyc, int iters, double a0) {
  acc_t acc[NACC];
  for (int e = 0; e < NACC; ++e) acc[e] = acc_t{ 0, 0, 0, 0 };
  double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[e], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int e = 0; e < NACC; ++e) s += acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// the same loop with a dependent fp64 FMA chain beside it (does the VALU run while the matrix pipe is busy?)
__global__ __launch_bounds__(256) void k_fma(double* out, unsigned long long* cyc, int iters, double a0) {
  double x[8];
  for (int e = 0; e < 8; ++e) x[e] = a0 + e + threadIdx.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fma(x[e], 1.0000001, 0.5);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int e = 0; e < 8; ++e) s += x[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
// One 512-thread workgroup per CU: waves 0-3 issue MFMAs, waves 4-7 fp64 FMAs, so every SIMD hosts one wave of each kind.
// Do the matrix pipe and the vector pipe run fp64 side by side (as they do for bf16 MFMA + VALU), or do they share the fp64 units?
__global__ __launch_bounds__(512) void k_mixed(double* out, unsigned long long* cyc, int iters_mfma, int iters_fma, double a0) {
  const int wave = threadIdx.x >> 6;
  unsigned long long t0, t1;
  double s = 0;
  if (wave < 4) {
    acc_t acc[8];
    for (int e = 0; e < 8; ++e) acc[e] = acc_t{ 0, 0, 0, 0 };
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[e], 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    for (int e = 0; e < 8; ++e) s += acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
  } else {
    double x[8];
    for (int e = 0; e < 8; ++e) x[e] = a0 + e + threadIdx.x;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = fma(x[e], 1.0000001, 0.5);
    }
    t1 = __builtin_readcyclecounter();
    for (int e = 0; e < 8; ++e) s += x[e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 8 * 2048 * 256 * 8); hipMalloc(&cyc, 8 * 2048 * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int threads : { 64, 128, 256 }) for (int bpc : { 1, 2, 4 }) {
    const int blocks = 256 * bpc;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8;
    const double waves = (double)blocks * threads / 64;
    std::printf("mfma_f64_16x16x4: %3d threads x %d blocks/CU: %.1f clock ticks per MFMA per wave, %.3f ms, %.1f TFLOP/s chip\n", threads, bpc,
                (double)c / nm, ms, waves * nm * 2048.0 / (ms * 1e-3) / 1e12);
  }
  for (int threads : { 64, 256 }) for (int bpc : { 1, 2, 3, 4, 6, 8 }) {
    const int blocks = 256 * bpc;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nf = (double)iters * 8;
    const double waves = (double)blocks * threads / 64;
    std::printf("v_fma_f64:        %3d threads x %d blocks/CU: %.1f clock ticks per FMA per wave, %.3f ms, %.1f TFLOP/s chip\n", threads, bpc,
                (double)c / nf, ms, waves * nf * 128.0 / (ms * 1e-3) / 1e12);
  }
  {
    // equal duration for both kinds when alone: 2000 x 8 MFMAs ~ 2.3 M cycles, 33 000 x 8 FMAs ~ 2.3 M cycles
    const int im = 2000, iff = 33000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mixed, dim3(256), dim3(512), 0, 0, out, cyc, im, iff, 1.0);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    std::printf("mixed, one MFMA wave + one FMA wave per SIMD: %.1f ticks per MFMA (alone 141.6), %.1f ticks per FMA (alone 8.5), %.3f ms; "
                "%.1f TFLOP/s chip (MFMA %.1f + FMA %.1f)\n", (double)c[0] / (im * 8.0), (double)c[4] / (iff * 8.0), ms,
                (1024.0 * im * 8 * 2048.0 + 1024.0 * iff * 8 * 128.0) / (ms * 1e-3) / 1e12,
                1024.0 * im * 8 * 2048.0 / ((double)c[0] / 2.4e9) / 1e12, 1024.0 * iff * 8 * 128.0 / ((double)c[4] / 2.4e9) / 1e12);
  }
  return 0;
}
