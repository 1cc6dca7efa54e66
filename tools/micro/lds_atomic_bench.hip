// Micro-benchmark: cost of LDS atomics per wave-instruction on gfx950 (ds_add_f64 vs ds_add_u64 vs ds_add_u32 vs plain
// read-modify-write), 8 one-wave workgroups per CU as in k_linearise_schur, addresses either distinct per lane or with
// the ~5-way same-address pattern of the camera records.   hipcc --offload-arch=gfx950 -O3 lds_atomic_bench.hip -o lab
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int CONFLICT>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
  __shared__ double S[2048];
  const int lane = threadIdx.x;
  for (int q = lane; q < 2048; q += 64) S[q] = 0.0;
  __syncthreads();
  // CONFLICT 1: 12 "cameras" -> ~5 lanes per address, spread over the wave; 2: 4 lanes per address, one in each 16-lane
  // row; 3: 4 adjacent lanes per address; 4: 2 lanes per address, 32 lanes apart; 5: 2 adjacent lanes per address
  const int base = CONFLICT == 1 ? (lane % 12) * 39 : CONFLICT == 2 ? (lane & 15) * 39 : CONFLICT == 3 ? (lane >> 2) * 39
                 : CONFLICT == 4 ? (lane & 31) * 39 : CONFLICT == 5 ? (lane >> 1) * 39 : lane * 31;
  double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 32; ++e) {
      const int a = (base + e) & 2047;
      if (MODE == 0) __hip_atomic_fetch_add(&S[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&S[a]), (unsigned long long)(long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(&S[a]), (unsigned int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 3) S[a] += v;
      if (MODE == 4) __hip_atomic_fetch_add(reinterpret_cast<float*>(&S[a]), (float)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    v += 1.0;
  }
  __syncthreads();
  double s = 0; for (int q = lane; q < 2048; q += 64) s += S[q];
  out[blockIdx.x * 64 + lane] = s;
}
template <int MODE, int CONFLICT> void run(const char* name) {
  const int blocks = 256 * 8, iters = 2000;
  double* out; hipMalloc(&out, blocks * 64 * sizeof(double));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, CONFLICT><<<blocks, 64>>>(out, 10);
  hipEventRecord(a); k<MODE, CONFLICT><<<blocks, 64>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 8 waves x iters x 32 instructions through one LDS
  const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * iters * 32);
  printf("%-28s conflict=%d  %.3f ms  -> %.1f LDS cycles per wave-instruction (at 2.4 GHz, 8 waves per CU)\n", name, CONFLICT, ms, cyc);
  hipFree(out);
}
int main() {
  run<0, 0>("ds_add_f64"); run<0, 1>("ds_add_f64");
  run<0, 2>("ds_add_f64"); run<0, 3>("ds_add_f64"); run<0, 4>("ds_add_f64"); run<0, 5>("ds_add_f64");
  run<1, 0>("ds_add_u64"); run<1, 1>("ds_add_u64");
  run<2, 0>("ds_add_u32"); run<2, 1>("ds_add_u32");
  run<4, 0>("ds_add_f32"); run<4, 1>("ds_add_f32");
  run<3, 0>("read + add + write (b64)");
  return 0;
}
