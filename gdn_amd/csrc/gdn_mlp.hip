// OutLayer MLP for out_layer_num > 1 (models/GDN.py:27-56,183) on the 16-bit matrix cores, eval mode:
//   h -> [Linear(K -> H), BatchNorm1d(H) (running statistics), ReLU] x (L-1) -> Linear(H -> 1)
// over the BN = batch*n rows of the head's output h2[BN, d] (gdn_head_fwd).
//
// The whole chain of one 32-row block lives in ONE wave's registers.  Every layer is computed
// TRANSPOSED, Y^T[H, 32 rows] = W' . A^T, with the row on the lane and the feature on the registers, so a
// layer's accumulators (+ bias, ReLU, split into two f16 terms) ARE the B operand of the next layer's
// product (sum over the feature = the accumulator's register index): no activation ever touches LDS or
// HBM.  fp32 accuracy comes from the same two-term f16 split as gdn_forward_dense.hip (hi*hi + lo*hi +
// hi*lo, fp32 accumulate).  The eval BatchNorm is folded into the weights (W' = scale W, b' = scale b +
// shift) when the PLAN is built (once per parameter update): the plan holds the split, folded weights in
// the order the kernel streams them — one 16-feature slab of every output row per step, staged through
// LDS for the 4 waves of a workgroup (double buffered).
//
// k order: a layer fed from accumulators sees feature 16s + 8(j>>2) + 4h + (j&3) in k slot 8h + j of
// step s (accumulator row map of v_mfma_f32_32x32x*), so the plan stores those layers' weight columns
// with bits 2 and 3 of the feature index exchanged; the first layer (fed from memory) is stored in
// natural order.
#include "gdn_common.hpp"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  // (a, b pinned as materialised fp32 values by two EMPTY asm statements: a product feeding the conversion is
  // otherwise contracted into v_fma_mixlo_f16 — f16 of the EXACT product — while the residual is taken against the
  // fp32-rounded one; see Fmt<FMT_F16>::pk in gdn_forward_dense.hip)
  asm("" : "+v"(a));
  asm("" : "+v"(b));
  const h2 p = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float neg_one_sgpr() {   // see gdn_forward_dense.hip: lets isel pick v_fma_mix
  float v = -1.0f;
  asm("" : "+s"(v));
  return v;
}
__device__ __forceinline__ void split8_f16(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned p = pk_f16(v[2 * j], v[2 * j + 1]);
    const h2 ph = __builtin_bit_cast(h2, p);
    hi[j] = p;
    lo[j] = pk_f16(__builtin_fmaf((float)ph[0], neg_one_sgpr(), v[2 * j]),
                   __builtin_fmaf((float)ph[1], neg_one_sgpr(), v[2 * j + 1]));
  }
}
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// ---- plan layout (bytes), shared by host and device -------------------------------------------------
// layer l (0 .. hidden_layers-1): KS_l slabs of [2 planes][Np rows][16 halfs], then bias'[Np] fp32;
// after the last hidden layer: w_out[Np] fp32, b_out fp32 (padded to 16 bytes).
struct MlpGeo {
  int d_in, hidden, hidden_layers;   // hidden_layers = out_layer_num - 1 >= 1
  int np, ks0, ksh;                  // Np = hidden rounded up to 32; k-steps of layer 0 / of the others
};
__host__ __device__ inline MlpGeo mlp_geo(int d_in, int hidden, int layers) {
  MlpGeo g;
  g.d_in = d_in; g.hidden = hidden; g.hidden_layers = layers - 1;
  g.np = (hidden + 31) & ~31;
  g.ks0 = (d_in + 15) / 16;
  g.ksh = g.np / 16;
  return g;
}
__host__ __device__ inline size_t mlp_slab_bytes(const MlpGeo& g) { return (size_t)g.np * 64; }
__host__ __device__ inline size_t mlp_layer_offset(const MlpGeo& g, int l) {
  size_t off = 0;
  for (int i = 0; i < l; ++i) off += (size_t)(i == 0 ? g.ks0 : g.ksh) * mlp_slab_bytes(g) + (size_t)g.np * 4;
  return off;
}
__host__ __device__ inline size_t mlp_plan_bytes(const MlpGeo& g) {
  return mlp_layer_offset(g, g.hidden_layers) + (size_t)g.np * 4 + 16;
}

// one hidden layer -> plan: fold BatchNorm, split into two f16 terms, slab order (+ k permutation)
__global__ __launch_bounds__(256) void mlp_plan_layer_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                             const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                             float eps, MlpGeo g, int layer, char* __restrict__ plan) {
  const int k_in = layer == 0 ? g.d_in : g.hidden;
  const int ks_n = layer == 0 ? g.ks0 : g.ksh;
  char* base = plan + mlp_layer_offset(g, layer);
  const int total = ks_n * g.np * 16;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int ks = t / (g.np * 16), rem = t - ks * g.np * 16;
    const int n = rem >> 4, slot = rem & 15;
    // feature held by k slot `slot` of step ks: natural for the first layer, accumulator order otherwise
    const int f16i = layer == 0 ? slot : ((slot & ~12) | ((slot & 4) << 1) | ((slot & 8) >> 1));
    const int k = ks * 16 + f16i;
    float v = 0.f;
    if (n < g.hidden && k < k_in) {
      const float scale = bn_w[n] / sqrtf(bn_var[n] + eps);
      v = w[(size_t)n * k_in + k] * scale;
    }
    asm("" : "+v"(v));          // the fp32-ROUNDED product is what gets split (see pk_f16)
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    _Float16* slab = reinterpret_cast<_Float16*>(base + (size_t)ks * mlp_slab_bytes(g));
    slab[(size_t)n * 16 + slot] = hi;
    slab[(size_t)g.np * 16 + (size_t)n * 16 + slot] = lo;
  }
  float* bias = reinterpret_cast<float*>(base + (size_t)ks_n * mlp_slab_bytes(g));
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < g.np; n += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (n < g.hidden) {
      const float scale = bn_w[n] / sqrtf(bn_var[n] + eps);
      v = fmaf(b[n], scale, bn_b[n] - bn_mean[n] * scale);
    }
    bias[n] = v;
  }
}

__global__ void mlp_plan_out_kernel(const float* __restrict__ w_out, const float* __restrict__ b_out, MlpGeo g,
                                    char* __restrict__ plan) {
  float* dst = reinterpret_cast<float*>(plan + mlp_layer_offset(g, g.hidden_layers));
  for (int n = threadIdx.x; n < g.np; n += blockDim.x) dst[n] = n < g.hidden ? w_out[n] : 0.f;
  if (threadIdx.x == 0) dst[g.np] = b_out[0];
}

// ---- the chain -----------------------------------------------------------------------------------------
// NT = Np / 32 output tiles per hidden layer; KS0 = k-steps of the first layer.
template <int NT, int KS0>
__global__ __launch_bounds__(256, 1) void mlp_fwd_kernel(const float* __restrict__ h2, const char* __restrict__ plan,
                                                         MlpGeo g, int rows, float* __restrict__ out) {
  constexpr int NP = 32 * NT, KSH = 2 * NT;
  constexpr int ROWB = 48;                       // LDS bytes per weight row of one plane (32 data + 16: conflict-free b128)
  constexpr int PLANE = NP * ROWB, SLAB_LDS = 2 * PLANE;
  __shared__ uint4 lds_u4[2 * SLAB_LDS / 16];
  char* lds = reinterpret_cast<char*>(lds_u4);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const size_t slab_bytes = (size_t)NP * 64;
  constexpr int PIECES = NP * 4;                 // 16-byte pieces of one slab (2 planes x NP rows x 2)

  // global -> registers -> LDS copy of one slab (PIECES / 256 pieces per thread)
  constexpr int PPT = (PIECES + 255) / 256;
  uint4 stage[PPT];
  auto slab_load = [&](const char* src) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int p = tid + u * 256;
      stage[u] = p < PIECES ? reinterpret_cast<const uint4*>(src)[p] : make_uint4(0, 0, 0, 0);
    }
  };
  auto slab_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int p = tid + u * 256;
      if (p < PIECES) {
        const int plane = p / (2 * NP), rem = p - plane * 2 * NP;
        *reinterpret_cast<uint4*>(lds + buf * SLAB_LDS + plane * PLANE + (rem >> 1) * ROWB + (rem & 1) * 16) = stage[u];
      }
    }
  };
  const int a_off = l32 * ROWB + h * 16;         // this lane's weight operand inside a tile of a plane

  for (int blk = blockIdx.x; blk * 128 < rows; blk += gridDim.x) {
    const int m = blk * 128 + wv * 32 + l32;     // this lane's row (as an MFMA column)
    const bool live = m < rows;
    // ---- first layer's B operand: the row's features in natural order
    u32x4 bh[KSH > KS0 ? KSH : KS0], bl[KSH > KS0 ? KSH : KS0];
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * h + j;
        const bool ok = live && k < g.d_in;
        const float t = h2[ok ? (size_t)m * g.d_in + k : 0];
        v[j] = ok ? t : 0.f;
      }
      split8_f16(v, bh[s], bl[s]);
    }
    f32x16 acc[NT];
    for (int layer = 0; layer < g.hidden_layers; ++layer) {
      const char* lbase = plan + mlp_layer_offset(g, layer);
      const int ks_n = layer == 0 ? KS0 : KSH;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      __syncthreads();                              // previous layer / block done with both LDS buffers
      slab_load(lbase);
      slab_store(0);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < (KSH > KS0 ? KSH : KS0); ++ks) {
        if (ks < ks_n) {
          if (ks + 1 < ks_n) slab_load(lbase + (size_t)(ks + 1) * slab_bytes);
          const char* sl = lds + (ks & 1) * SLAB_LDS;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const u32x4 ah = *reinterpret_cast<const u32x4*>(sl + t * 32 * ROWB + a_off);
            const u32x4 al = *reinterpret_cast<const u32x4*>(sl + PLANE + t * 32 * ROWB + a_off);
            acc[t] = mfma16(ah, bh[ks], acc[t]);
            acc[t] = mfma16(al, bh[ks], acc[t]);
            acc[t] = mfma16(ah, bl[ks], acc[t]);
          }
          if (ks + 1 < ks_n) slab_store((ks + 1) & 1);
          __syncthreads();
        }
      }
      // bias' + ReLU; the accumulators become the next layer's B operand (k-steps 2t, 2t+1 of tile t)
      const float* bias = reinterpret_cast<const float*>(lbase + (size_t)ks_n * slab_bytes);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(bias + 32 * t + 8 * q + 4 * h);
          a[4 * q] = fmaxf(acc[t][4 * q] + b4.x, 0.f);
          a[4 * q + 1] = fmaxf(acc[t][4 * q + 1] + b4.y, 0.f);
          a[4 * q + 2] = fmaxf(acc[t][4 * q + 2] + b4.z, 0.f);
          a[4 * q + 3] = fmaxf(acc[t][4 * q + 3] + b4.w, 0.f);
        }
        if (layer + 1 < g.hidden_layers) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = a[8 * s + j];
            split8_f16(v, bh[2 * t + s], bl[2 * t + s]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = a[r];
        }
      }
    }
    // ---- Linear(H -> 1): lane local over the registers, then the two lane halves
    const float* wo = reinterpret_cast<const float*>(plan + mlp_layer_offset(g, g.hidden_layers));
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w4 = *reinterpret_cast<const float4*>(wo + 32 * t + 8 * q + 4 * h);
        part = fmaf(acc[t][4 * q], w4.x, part);
        part = fmaf(acc[t][4 * q + 1], w4.y, part);
        part = fmaf(acc[t][4 * q + 2], w4.z, part);
        part = fmaf(acc[t][4 * q + 3], w4.w, part);
      }
    part += __shfl_xor(part, 32);
    if (h == 0 && live) out[m] = part + wo[NP];
  }
}

template <int NT>
int mlp_launch_ks0(const float* h2, const char* plan, const MlpGeo& g, int rows, float* out, hipStream_t st) {
  const int grid = max(1, min((rows + 127) / 128, gdn_cu_count() * 2));
#define GDN_MLP_CASE(K) \
  case K: hipLaunchKernelGGL((mlp_fwd_kernel<NT, K>), dim3(grid), dim3(256), 0, st, h2, plan, g, rows, out); break;
  switch (g.ks0) {
    GDN_MLP_CASE(1) GDN_MLP_CASE(2) GDN_MLP_CASE(4) GDN_MLP_CASE(8)
    default: return GDN_ERR_UNSUPPORTED;
  }
#undef GDN_MLP_CASE
  return gdn_launch_status();
}

bool mlp_supported(int d_in, int hidden, int layers) {
  return layers >= 2 && layers <= 8 && hidden >= 1 && hidden <= 256 &&
         (d_in == 16 || d_in == 32 || d_in == 64 || d_in == 128);
}

}  // namespace

extern "C" long long gdn_mlp_plan_bytes(int d_in, int hidden, int layers) {
  if (!mlp_supported(d_in, hidden, layers)) return 0;
  return (long long)mlp_plan_bytes(mlp_geo(d_in, hidden, layers));
}

extern "C" int gdn_mlp_plan_layer(const float* weight, const float* bias, const float* bn_weight,
                                  const float* bn_bias, const float* bn_mean, const float* bn_var, float eps,
                                  int d_in, int hidden, int layers, int layer, void* plan, void* stream) {
  if (!weight || !bias || !bn_weight || !bn_bias || !bn_mean || !bn_var || !plan) return GDN_ERR_ARG;
  if (!mlp_supported(d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  if (layer < 0 || layer >= layers - 1) return GDN_ERR_ARG;
  const MlpGeo g = mlp_geo(d_in, hidden, layers);
  hipLaunchKernelGGL(mlp_plan_layer_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, weight, bias, bn_weight,
                     bn_bias, bn_mean, bn_var, eps, g, layer, reinterpret_cast<char*>(plan));
  return gdn_launch_status();
}

extern "C" int gdn_mlp_plan_out(const float* weight, const float* bias, int d_in, int hidden, int layers,
                                void* plan, void* stream) {
  if (!weight || !bias || !plan) return GDN_ERR_ARG;
  if (!mlp_supported(d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  const MlpGeo g = mlp_geo(d_in, hidden, layers);
  hipLaunchKernelGGL(mlp_plan_out_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weight, bias, g,
                     reinterpret_cast<char*>(plan));
  return gdn_launch_status();
}

extern "C" int gdn_mlp_fwd(const float* h2, const void* plan, int rows, int d_in, int hidden, int layers,
                           float* out, void* stream) {
  if (!h2 || !plan || !out || rows <= 0) return GDN_ERR_ARG;
  if (!mlp_supported(d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  const MlpGeo g = mlp_geo(d_in, hidden, layers);
  const char* p = reinterpret_cast<const char*>(plan);
  hipStream_t st = (hipStream_t)stream;
  switch (g.np / 32) {
    case 1: return mlp_launch_ks0<1>(h2, p, g, rows, out, st);
    case 2: return mlp_launch_ks0<2>(h2, p, g, rows, out, st);
    case 3: return mlp_launch_ks0<3>(h2, p, g, rows, out, st);
    case 4: return mlp_launch_ks0<4>(h2, p, g, rows, out, st);
    case 5: return mlp_launch_ks0<5>(h2, p, g, rows, out, st);
    case 6: return mlp_launch_ks0<6>(h2, p, g, rows, out, st);
    case 7: return mlp_launch_ks0<7>(h2, p, g, rows, out, st);
    case 8: return mlp_launch_ks0<8>(h2, p, g, rows, out, st);
  }
  return GDN_ERR_UNSUPPORTED;
}
