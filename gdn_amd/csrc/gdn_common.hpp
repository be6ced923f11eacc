// Shared device helpers for the gfx950 GDN kernels.  wave = 64 lanes, DPP "row" = 16 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/gdn_hip.h"

#define GDN_NEG_SLOPE 0.2f      // LeakyReLU slope, reference models/graph_layer.py:13
#define GDN_SOFTMAX_EPS 1e-16f  // torch_geometric.utils.softmax 1.5.0 denominator epsilon
#define GDN_MAX_W 64
#define GDN_A_PITCH 64          // a_i / a_j are stored zero padded to 64 floats

static inline int gdn_launch_status() {
  return hipGetLastError() == hipSuccess ? GDN_OK : GDN_ERR_LAUNCH;
}

// Immutable device properties, cached per device behind a mutex (gdn_graph.hip): the library is re-entrant
// across threads, streams and devices.  gdn_blocks_per_cu = resident workgroups per CU of kernel `fn` at
// (threads, dynamic LDS bytes); it also raises the kernel's dynamic-LDS limit to the 160 KB of a gfx950 CU.
int gdn_cu_count();
int gdn_blocks_per_cu(const void* fn, int threads, int lds);
// environment knob of a diagnostic A/B run, read ONCE per process at its first use (a function-local static:
// thread-safe initialisation, nothing on the launch path afterwards)
#define GDN_ENV_INT_ONCE(NAME, FALLBACK) \
  ([]() -> int { static const int v = [] { const char* e = getenv(NAME); return e ? atoi(e) : (FALLBACK); }(); return v; }())

int gdn_forward_staged_ok(int n, int w, int d, int k);        // gdn_forward.hip: the staged forward takes the shape
// gdn_forward_dense.hip: the matrix-core aggregation path (n <= 127, d = 64); x is fp32 or bf16 bits
bool gdn_dense_supported(int n, int w, int d, int k);         // staged kernels: d = 64
bool gdn_dense_fused_supported(int n, int w, int d, int k);   // fused kernel: d = 64 or 128
int gdn_dense_forward_fused(const void* x, int x_is_bf16, int series_len, int series_first, const float* lin_w,
                            const float* node_terms, const uint16_t* nbr, const float* gnn_bias,
                            const float* emb, const float* bn1, const float* bn2, const float* out_w,
                            const float* out_b, int batch, int n, int w, int d, int k, float* out,
                            hipStream_t stream);
int gdn_dense_attn_aggregate(const void* xlin, int is_bf16, const float* s_i, const float* s_j, const uint16_t* nbr,
                             const float* bias, int batch, int n, int d, int k, void* z, float* alpha,
                             hipStream_t stream);
int gdn_dense_attn_bwd(const float* d_z, const float* xlin, const float* alpha, const float* s_i, const float* s_j,
                       const uint16_t* nbr, int batch, int n, int k, float* d_xlin, float* d_si, float* d_sj,
                       float* d_bias, float* bias_ws, hipStream_t stream);   // matrix-core backward of the gather-aggregate
int gdn_dense_project(const void* x, int is_bf16, const float* lin_w, const float* node_terms, int batch, int n,
                      int w, int d, void* xlin, float* s_i, float* s_j, hipStream_t stream);
// run-time choice between the two fused forward implementations: GDN_FUSED_PATH=valu keeps the fp32 VALU
// row-gather kernel for every shape (read once per process)
static inline bool gdn_use_dense_path() {
  static const int use = [] {
    const char* e = getenv("GDN_FUSED_PATH");
    return (e && e[0] == 'v') ? 0 : 1;
  }();
  return use != 0;
}

// ---- DPP within a 16-lane row ------------------------------------------------------
// dpp_ctrl encodings (gfx9): quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_mirror 0x140,
// row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
#define GDN_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define GDN_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define GDN_DPP_HALF_MIRROR 0x141  // lane i <-> 7-i inside each 8
#define GDN_DPP_MIRROR 0x140       // lane i <-> 15-i inside each 16
#define GDN_DPP_ROR1 0x121         // rotate the 16-lane row by one lane

// All 16 lanes of a row end with the row's sum / max.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<GDN_DPP_XOR1>(v);
  v += dpp_f<GDN_DPP_XOR2>(v);
  v += dpp_f<GDN_DPP_HALF_MIRROR>(v);
  v += dpp_f<GDN_DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<GDN_DPP_XOR1>(v));
  v = fmaxf(v, dpp_f<GDN_DPP_XOR2>(v));
  v = fmaxf(v, dpp_f<GDN_DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<GDN_DPP_MIRROR>(v));
  return v;
}

// LeakyReLU(0.2): max(v, 0.2 v) — v for v > 0, 0.2 v otherwise; -inf stays -inf
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, v * GDN_NEG_SLOPE); }

// 64-lane wave sum through DPP + readlane (used by the small reduction kernels).
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Fixed-order column sum across the workgroups of a launch, no floating-point atomics: every workgroup stores its
// row of `cols` partial sums (`lds_row`, complete and visible to the workgroup: call after a barrier), takes a
// ticket, and the workgroup that draws the last one adds the rows IN ROW ORDER and writes out[cols].  The result
// depends on the grid size only, never on the order the workgroups finish in: bitwise reproducible.
//   ws: [0..3] = the ticket (u64 + padding; zero on entry, left zero), rows at ws + 4 + row * cols.
// Same memory protocol as gdn_mse_kernel: agent-scope stores acknowledged (vmcnt) before the ticket is drawn, the
// last workgroup reads the rows back with agent-scope loads (no release/acquire fence: each would write back an
// XCD's whole L2).  Must be reached by every thread of every workgroup; `scratch` = blockDim.x floats of LDS.
#define GDN_COLSUM_WS_HEAD 4
#define GDN_COLSUM_MAX_ROWS 1024   // launches that use it keep their grid at or below this
__device__ __forceinline__ void gdn_colsum_ticket(float* __restrict__ ws, const float* lds_row, int cols,
                                                  float* __restrict__ out, float* scratch) {
  __shared__ bool last_wg;
  const int tid = threadIdx.x, nth = blockDim.x;
  float* rows = ws + GDN_COLSUM_WS_HEAD;
  for (int c = tid; c < cols; c += nth)
    __hip_atomic_store(rows + (size_t)blockIdx.x * cols + c, lds_row[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(ws);
    const unsigned long long t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_wg = t == (unsigned long long)gridDim.x - 1ull;
    if (last_wg) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last_wg) return;
  // lane groups take rows g, g + groups, ...; four independent chains each so the loads overlap
  const int groups = nth / cols > 0 ? nth / cols : 1;
  const int c = tid % cols, g = tid / cols;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (g < groups) {
    const int nrow = (int)gridDim.x;
    int r = g;
    for (; r + 3 * groups < nrow; r += 4 * groups) {
      s0 += __hip_atomic_load(rows + (size_t)r * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s1 += __hip_atomic_load(rows + (size_t)(r + groups) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(rows + (size_t)(r + 2 * groups) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s3 += __hip_atomic_load(rows + (size_t)(r + 3 * groups) * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (; r < nrow; r += groups)
      s0 += __hip_atomic_load(rows + (size_t)r * cols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  scratch[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tid < cols) {
    float s = 0.f;
    for (int q = 0; q < groups; ++q) s += scratch[q * cols + tid];
    out[tid] = s;
  }
}

// Sum of the per-workgroup partial rows of gdn_project_bwd ([rows][D*wp + 128 + 2n] -> d_lin_w[D,w], d_a[2,64],
// d_c[2,n]): workgroup `block` sums 16 columns — 16 lane groups take every 16th row (four independent chains
// each, so the loads overlap), then an LDS reduce.  A device function because two kernels run it: the plain
// reduce launch and the training step's combined tail launch (gdn_head_train.hip).
__device__ __forceinline__ void gdn_project_reduce_body(const float* __restrict__ part, int rows, int d, int n, int w,
                                                        int wp, float* __restrict__ d_lin_w, float* __restrict__ d_a,
                                                        float* __restrict__ d_c, int block) {
  __shared__ float red[16][17];
  const int len = d * wp + 128 + 2 * n;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int t = block * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (t < len) {
    int r = g;
    for (; r + 48 < rows; r += 64) {
      s0 += part[(size_t)r * len + t];
      s1 += part[(size_t)(r + 16) * len + t];
      s2 += part[(size_t)(r + 32) * len + t];
      s3 += part[(size_t)(r + 48) * len + t];
    }
    for (; r < rows; r += 16) s0 += part[(size_t)r * len + t];
  }
  red[g][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g != 0 || t >= len) return;
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += red[q][c];
  if (t < d * wp) {
    const int cc = t / d, r = t - cc * d;     // partial rows hold d_lin_w column-major (c*D + d)
    if (cc < w) d_lin_w[(size_t)r * w + cc] = s;
  } else if (t < d * wp + 128) {
    d_a[t - d * wp] = s;
  } else {
    d_c[t - d * wp - 128] = s;
  }
}
