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

static inline int gdn_cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

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
                       float* d_bias, hipStream_t stream);        // matrix-core backward of the gather-aggregate
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
