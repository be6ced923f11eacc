// Micro-probe: cost of LDS atomics in a 1024-thread workgroup (the select kernel's histogram passes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const unsigned* __restrict__ digits, unsigned* out, long long* ticks) {
  __shared__ unsigned hist[16][256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 16 * 256; i += 1024) (&hist[0][0])[i] = 0;
  unsigned d[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) d[u] = digits[(blockIdx.x * 32 + u) * 1024 + tid];
  __syncthreads();
  const long long t0 = wall_clock64();
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    if (MODE == 0) { if (lane == 0) atomicAdd(&hist[0][d[u] & 0], 64u); }            // lane 0, one address
    if (MODE == 1) atomicAdd(&hist[0][d[u]], 1u);                                      // all lanes, shared hist
    if (MODE == 2) atomicAdd(&hist[wv][d[u]], 1u);                                     // all lanes, per-wave hist
    if (MODE == 3) { if (lane == 0) atomicAdd(&hist[wv][0], 64u); }                    // lane 0, per-wave address
    if (MODE == 4) atomicAdd(&hist[0][(d[u] & 0xf0) | (lane & 15)], 1u);               // few conflicts
    if (MODE == 5) hist[wv][(lane * 4 + u) & 255] += 1;                                // plain RMW (no atomics)
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  unsigned s = 0;
  for (int i = tid; i < 16 * 256; i += 1024) s += (&hist[0][0])[i];
  out[blockIdx.x * 1024 + tid] = s;
}

int main() {
  const int G = 127;
  unsigned* h = (unsigned*)malloc(G * 32 * 1024 * 4);
  srand(1);
  for (int i = 0; i < G * 32 * 1024; ++i) {
    // triangular-ish digit distribution over ~80 bins like |a-b| second byte
    const double x = fabs((double)rand() / RAND_MAX - (double)rand() / RAND_MAX);
    unsigned long long bits; memcpy(&bits, &x, 8);
    h[i] = (unsigned)(bits >> 48) & 255u;
  }
  unsigned *d, *out; long long* ticks;
  hipMalloc(&d, G * 32 * 1024 * 4); hipMalloc(&out, G * 1024 * 4); hipMalloc(&ticks, G * 8);
  hipMemcpy(d, h, G * 32 * 1024 * 4, hipMemcpyHostToDevice);
  long long ht[G];
#define RUN(M) for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(probe<M>, dim3(G), dim3(1024), 0, 0, d, out, ticks); hipDeviceSynchronize(); } \
  hipMemcpy(ht, ticks, G * 8, hipMemcpyDeviceToHost); printf("mode %d: %lld ticks(10ns) for 32 rounds x 16 waves\n", M, ht[5]);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
