// tools/micro/zero_copy_bench.hip - what a STREAM of windows can move over the host link without a host thread touching the data:
//   (a) hipMemcpyAsync from pinned memory (the copy engine), one large copy and many window-sized ones (per-call host cost),
//   (b) a kernel that reads pinned host memory directly (zero copy) with few workgroups - the ingest of the device-built refill,
//   (c) a kernel that writes results straight into pinned host memory,
//   (d) (b) and (c) at once (the link is full duplex),
//   (e) hipHostRegister of a caller's pageable buffer (per-call cost).
// hipcc --offload-arch=gfx950 -O3 tools/micro/zero_copy_bench.hip -o /tmp/zcb && /tmp/zcb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// every thread moves 16 bytes per step, `U` independent loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) dst[i + u * 256] = v[u];
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t bytes = (size_t)1 << 30;
  const long long n16 = (long long)(bytes / 16);
  double2 *h_in = nullptr, *h_out = nullptr, *d_a = nullptr, *d_b = nullptr;
  CK(hipHostMalloc((void**)&h_in, bytes, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&h_out, bytes, hipHostMallocDefault));
  CK(hipMalloc((void**)&d_a, bytes));
  CK(hipMalloc((void**)&d_b, bytes));
  std::memset(h_in, 1, bytes); std::memset(h_out, 0, bytes);
  CK(hipMemset(d_b, 3, bytes));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](const char* what, size_t moved, auto fn) {
    fn(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0, s)); fn(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::printf("%-64s %8.2f ms  %7.2f GB/s\n", what, best, moved / best / 1e6);
  };
  timed("(a) hipMemcpyAsync H2D pinned, 1 GiB in one copy", bytes, [&] { CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s)); });
  timed("(a) hipMemcpyAsync D2H pinned, 1 GiB in one copy", bytes, [&] { CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s)); });
  for (size_t piece : { (size_t)64 << 10, (size_t)800 << 10 }) {
    const size_t np = bytes / piece;
    double host_s = 0;
    char label[128];
    std::snprintf(label, sizeof label, "(a) H2D pinned as %zu copies of %zu KiB", np, piece >> 10);
    timed(label, bytes, [&] { const double t = now(); for (size_t q = 0; q < np; ++q) CK(hipMemcpyAsync((char*)d_a + q * piece, (char*)h_in + q * piece, piece, hipMemcpyHostToDevice, s)); host_s = now() - t; });
    std::printf("     host time to enqueue: %.2f ms = %.2f us per call\n", host_s * 1e3, host_s * 1e6 / np);
  }
  for (int wgs : { 8, 16, 32, 64, 128, 256, 1024 }) {
    char label[128];
    std::snprintf(label, sizeof label, "(b) kernel reads pinned host memory, %d workgroups x 256, 4 x 16 B", wgs);
    timed(label, bytes, [&] { hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, s, (const double2*)h_in, d_a, n16); });
  }
  for (int wgs : { 16, 64 }) {
    char label[128];
    std::snprintf(label, sizeof label, "(b) kernel reads pinned host memory, %d workgroups x 256, 8 x 16 B", wgs);
    timed(label, bytes, [&] { hipLaunchKernelGGL(k_copy<8>, dim3(wgs), dim3(256), 0, s, (const double2*)h_in, d_a, n16); });
  }
  for (int wgs : { 16, 64, 256 }) {
    char label[128];
    std::snprintf(label, sizeof label, "(c) kernel writes pinned host memory, %d workgroups x 256", wgs);
    timed(label, bytes, [&] { hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, s, (const double2*)d_b, h_out, n16); });
  }
  timed("(d) kernel read (64 wg) and kernel write (64 wg) at once, 2 GiB moved", 2 * bytes, [&] {
    hipLaunchKernelGGL(k_copy<4>, dim3(64), dim3(256), 0, s2, (const double2*)d_b, h_out, n16);
    hipLaunchKernelGGL(k_copy<4>, dim3(64), dim3(256), 0, s, (const double2*)h_in, d_a, n16);
    CK(hipStreamSynchronize(s2));
  });
  timed("(d) copy engine H2D and D2H at once, 2 GiB moved", 2 * bytes, [&] {
    CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2));
    CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s2));
  });
  {
    // (e) registering a caller's pageable buffer: per-call cost against the copy it saves
    const size_t piece = (size_t)64 << 20;
    char* p = (char*)std::aligned_alloc(4096, piece);
    std::memset(p, 5, piece);
    double t = now();
    CK(hipHostRegister(p, piece, hipHostRegisterDefault));
    const double reg = now() - t;
    t = now();
    CK(hipHostUnregister(p));
    const double unreg = now() - t;
    std::printf("(e) hipHostRegister of 64 MiB pageable: %.2f ms, unregister %.2f ms\n", reg * 1e3, unreg * 1e3);
    t = now();
    CK(hipMemcpy(d_a, p, piece, hipMemcpyHostToDevice));
    std::printf("(e) hipMemcpy H2D of the same 64 MiB, pageable: %.2f ms = %.2f GB/s\n", (now() - t) * 1e3, piece / (now() - t) / 1e9);
    std::free(p);
  }
  {
    // host memcpy into pinned memory, one thread: what a staged ingest costs per byte
    char* p = (char*)std::aligned_alloc(4096, bytes);
    std::memset(p, 7, bytes);
    double t = now();
    std::memcpy(h_in, p, bytes);
    std::printf("(f) one host thread memcpy pageable -> pinned, 1 GiB: %.2f ms = %.2f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    std::free(p);
  }
  return 0;
}
