#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// atomics throughput probe: N threads each do K atomic adds to pseudo-random (or coherent) addresses
template <int MODE>
__global__ void k_atom(float* tab, unsigned entries, int K, int coherent, unsigned xcd_stride) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned xcc = 0;
  if (MODE == 2 || MODE == 3) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
  }
  float* t = tab + (size_t)xcc * xcd_stride;
  unsigned s = coherent ? (tid / 16) * 2654435761u : tid * 2654435761u;
  for (int k = 0; k < K; ++k) {
    s = s * 1664525u + 1013904223u;
    unsigned idx = (s >> 8) % entries;
    float* p = t + (size_t)idx * 2;
    if (MODE == 0) { unsafeAtomicAdd(p, 1.0f); unsafeAtomicAdd(p + 1, 1.0f); }
    if (MODE == 1 || MODE == 2) {
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE == 3) {  // packed half2, workgroup scope, xcd private
      __half2* hp = (__half2*)p;
      unsafeAtomicAdd(hp, __half2(__float2half(1.0f), __float2half(1.0f)));
    }
    if (MODE == 4) {  // packed half2 agent scope
      __half2* hp = (__half2*)p;
      unsafeAtomicAdd(hp, __half2(__float2half(1.0f), __float2half(1.0f)));
    }
    if (MODE == 5) {  // plain (non atomic) RMW for reference
      p[0] += 1.0f; p[1] += 1.0f;
    }
  }
}
template <int MODE>
void run(const char* name, float* tab, unsigned entries, int coherent, unsigned stride) {
  const int threads = 262144 * 7, K = 32;  // 58.7M pair-adds (=117M float atomics for f32 modes)
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_atom<MODE><<<threads / 256, 256>>>(tab, entries, K, coherent, stride);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 3; ++r) k_atom<MODE><<<threads / 256, 256>>>(tab, entries, K, coherent, stride);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
  printf("%-44s entries=%8u coherent=%d : %8.3f ms  %7.2f G pair-adds/s\n", name, entries, coherent, ms, threads * (double)K / ms / 1e6);
}
int main() {
  const unsigned maxe = 4u << 20;
  float* tab; hipMalloc(&tab, (size_t)maxe * 2 * 4 * 8); hipMemset(tab, 0, (size_t)maxe * 2 * 4 * 8);
  for (unsigned entries : {32768u, 1u << 19, 1u << 21}) for (int coh : {0, 1}) {
    run<0>("f32 unsafeAtomicAdd agent", tab, entries, coh, 0);
    run<1>("f32 fetch_add workgroup scope (shared tab)", tab, entries, coh, 0);
    run<2>("f32 fetch_add workgroup scope, XCD-private", tab, entries, coh, maxe * 2);
    run<3>("half2 pk atomic, XCD-private", tab, entries, coh, maxe * 2);
    run<4>("half2 pk atomic agent", tab, entries, coh, 0);
    run<5>("plain RMW (non-atomic)", tab, entries, coh, 0);
  }
  return 0;
}
