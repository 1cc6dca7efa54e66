// tools/micro/mfma_f64_bench.hip — how fast is v_mfma_f64_16x16x4_f64 on gfx950?  Independent accumulators, one to eight
// waves per CU; prints cycles per MFMA per wave (s_memtime) and the chip-wide TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_bench tools/micro/mfma_f64_bench.hip && ./mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double acc_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, unsigned long long* cyc, int iters, double a0) {
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
  return 0;
}
