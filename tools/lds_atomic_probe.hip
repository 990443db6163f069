// LDS atomic cost probe: cycles per ds_add_f32 / ds_cmpst_rtn as a function of active lanes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(int active_lanes, int iters, float* out, long long* cyc) {
  __shared__ float vals[8192];
  __shared__ uint32_t keys[4096];
  for (int i = threadIdx.x; i < 8192; i += 256) vals[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) keys[i] = 0xFFFFFFFFu;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = lane < active_lanes;
  uint32_t a = (lane * 37 + wave * 1031) & 4095;
  long long t0 = clock64();
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (on) {
      if (MODE == 0) {            // non-returning float add
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&vals[(a + 64 * k) & 8191], 1.0f);
      } else if (MODE == 1) {     // returning CAS, 8 independent then use
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = atomicCAS(&keys[(a + 64 * k) & 4095], 0xFFFFFFFFu, a + k);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (float)(o[k] & 1);
      } else if (MODE == 2) {     // plain LDS store b32
#pragma unroll
        for (int k = 0; k < 8; ++k) vals[(a + 64 * k) & 8191] = (float)it;
      } else if (MODE == 4) {     // non-returning u32 add
        uint32_t* kv = keys;
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&kv[(a + 64 * k) & 4095], 3u);
      } else if (MODE == 5) {     // non-returning u64 add
        unsigned long long* kv = reinterpret_cast<unsigned long long*>(vals);
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&kv[(a + 64 * k) & 4095], 3ull);
      } else if (MODE == 3) {     // all lanes same address add
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&vals[k], 1.0f);
      }
    }
    a = (a + 17) & 4095;
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 123.f) out[0] = acc + vals[lane];
  out[1 + (threadIdx.x & 7)] = vals[threadIdx.x];
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  const char* names[6] = {"ds_add_f32 (8/iter)", "ds_cmpst_rtn x8 batched", "ds_write_b32 x8", "ds_add_f32 same addr", "ds_add_u32", "ds_add_u64"};
  for (int mode = 4; mode < 6; ++mode)
    for (int lanes : {1, 4, 16, 64}) {
      for (int blocks : {256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (mode == 0) probe<0><<<blocks, 256>>>(lanes, iters, out, cyc);
          if (mode == 1) probe<1><<<blocks, 256>>>(lanes, iters, out, cyc);
          if (mode == 2) probe<2><<<blocks, 256>>>(lanes, iters, out, cyc);
          if (mode == 3) probe<3><<<blocks, 256>>>(lanes, iters, out, cyc);
          if (mode == 4) probe<4><<<blocks, 256>>>(lanes, iters, out, cyc);
          if (mode == 5) probe<5><<<blocks, 256>>>(lanes, iters, out, cyc);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-26s lanes %2d: %.3f ms, %.1f clk/instr per wave (4 waves/CU share the LDS), s_memtime ticks/instr %.1f\n",
               names[mode], lanes, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0), (double)c / (iters * 8.0));
      }
    }
  return 0;
}
