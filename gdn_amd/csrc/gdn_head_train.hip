// Train-mode output head and its backward, out_layer_num == 1:
//
//   y1 = BN1(z)  a1 = relu(y1)            GNNLayer.bn + relu      (reference models/GDN.py:77-79)
//   h1 = a1 * emb[node]                   torch.mul(out, embedding)                 (:175-176)
//   y2 = BN2(h1) a2 = relu(y2)            bn_outlayer_in over [B, d, N] + relu      (:178-180)
//   out = sum_d a2 * mask * w[d] + b      dropout(0.2) + OutLayer Linear(d -> 1)    (:182-184)
//
// Both BatchNorms are in training mode: they normalise by the statistics of THIS batch (all B*N
// rows, biased variance) and update their running estimates (momentum, unbiased variance).
//
// Everything is recomputed from z in each pass, so no [B*N, d] intermediate is ever stored; a pass
// streams z (and the dropout mask) once and is HBM bound.  Forward = 3 passes (statistics of z,
// statistics of h1, output), backward = 3 passes (BN2 reductions + Linear gradients, BN1 reductions +
// embedding gradient, d_z).  Column reductions are carried in fp64 — per thread, across the
// workgroup through LDS and across workgroups with fp64 atomics — so the result does not depend on
// the accumulation order to fp32 precision.
//
// Thread layout: a row (one sensor of one window) is covered by LPR = d/4 consecutive lanes holding
// four columns each; a workgroup owns whole windows, and lane group `slot` always works on sensors
// slot, slot+SLOTS, ..., so per-(sensor, column) accumulators are private to a thread.
#include "gdn_common.hpp"

namespace {

template <int D>
struct HG {
  static constexpr int LPR = D / 4;
  static constexpr int SLOTS = 256 / LPR;
};

struct HeadArgs {
  const float *z, *emb, *g1, *b1, *g2, *b2, *w, *bo, *mask, *d_out;
  const double* fstats;  // [4][d]: sum z, sum z^2, sum h1, sum h1^2
  double* acc;           // forward passes: fstats (writable); backward: [6][d] + [n*d] workspace
  float *out, *d_z;
  int batch, n;
  float eps1, eps2;
  int demb_lds;          // the [n,d] embedding-gradient partial fits LDS
};

struct BnCols {
  float mu[4], is[4], sc[4], be[4];
};

__device__ __forceinline__ BnCols bn_cols(const double* sum, const double* sq, double rows, float eps,
                                          const float* gamma, const float* beta, int c0) {
  BnCols r;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const double m = sum[c0 + v] / rows;
    double var = sq[c0 + v] / rows - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    r.mu[v] = (float)m;
    r.is[v] = is;
    r.sc[v] = gamma[c0 + v] * is;
    r.be[v] = beta[c0 + v];
  }
  return r;
}

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// sum the per-thread column partials over the workgroup's SLOTS lane groups, then one fp64 atomic per
// column per workgroup
template <int D>
__device__ __forceinline__ void col_reduce(const double (&v)[4], double* red, double* gdst, int tid, int slot,
                                           int c0) {
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) red[slot * D + c0 + q] = v[q];
  __syncthreads();
  if (tid < D) {
    double s = 0.0;
    for (int q = 0; q < HG<D>::SLOTS; ++q) s += red[q * D + tid];
    atomicAdd(gdst + tid, s);
  }
}

enum { H_STAT1 = 0, H_STAT2 = 1, H_OUT = 2, H_BWD2 = 3, H_BWD1 = 4, H_DZ = 5 };

template <int D, int MODE>
__global__ __launch_bounds__(256) void gdn_head_train_kernel(const HeadArgs a) {
  using G = HG<D>;
  extern __shared__ double smem_d[];
  double* red = smem_d;                                       // [SLOTS][D] = 1024 doubles
  float* demb_l = reinterpret_cast<float*>(smem_d + 1024);    // [n][D] (H_BWD1, when it fits)
  const int tid = threadIdx.x, lr = tid % G::LPR, slot = tid / G::LPR, c0 = lr * 4;
  const double rows = (double)a.batch * (double)a.n;

  BnCols bn1 = {}, bn2 = {};
  if constexpr (MODE >= H_STAT2) bn1 = bn_cols(a.fstats, a.fstats + D, rows, a.eps1, a.g1, a.b1, c0);
  if constexpr (MODE >= H_OUT) bn2 = bn_cols(a.fstats + 2 * D, a.fstats + 3 * D, rows, a.eps2, a.g2, a.b2, c0);
  float w4[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (MODE >= H_OUT) ld4(a.w + c0, w4);
  float m2a[4] = {}, m2b[4] = {}, m1a[4] = {}, m1b[4] = {};
  if constexpr (MODE >= H_BWD1) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      m2a[v] = (float)(a.acc[c0 + v] / rows);           // mean of d_y2
      m2b[v] = (float)(a.acc[D + c0 + v] / rows);       // mean of d_y2 * xhat2
    }
  }
  if constexpr (MODE == H_DZ) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      m1a[v] = (float)(a.acc[2 * D + c0 + v] / rows);
      m1b[v] = (float)(a.acc[3 * D + c0 + v] / rows);
    }
  }
  double acc0[4] = {0.0, 0.0, 0.0, 0.0}, acc1[4] = {0.0, 0.0, 0.0, 0.0}, acc2[4] = {0.0, 0.0, 0.0, 0.0};
  double acc_s = 0.0;
  if constexpr (MODE == H_BWD1) {
    if (a.demb_lds) {
      for (int t = tid; t < a.n * D; t += 256) demb_l[t] = 0.f;
      __syncthreads();
    }
  }
  const float bias_o = (MODE == H_OUT) ? a.bo[0] : 0.f;

  for (int b = blockIdx.x; b < a.batch; b += gridDim.x) {
    for (int n = slot; n < a.n; n += G::SLOTS) {
      const size_t off = ((size_t)b * a.n + n) * D + c0;
      float z[4];
      ld4(a.z + off, z);
      if constexpr (MODE == H_STAT1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double zd = (double)z[v];
          acc0[v] += zd;
          acc1[v] = fma(zd, zd, acc1[v]);
        }
        continue;
      }
      float e[4], y1[4], a1[4], h1[4];
      ld4(a.emb + (size_t)n * D + c0, e);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        y1[v] = fmaf(z[v] - bn1.mu[v], bn1.sc[v], bn1.be[v]);
        a1[v] = fmaxf(y1[v], 0.f);
        h1[v] = a1[v] * e[v];
      }
      if constexpr (MODE == H_STAT2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const double hd = (double)h1[v];
          acc0[v] += hd;
          acc1[v] = fma(hd, hd, acc1[v]);
        }
        continue;
      }
      float y2[4], a2[4], m[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.mask) ld4(a.mask + off, m);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        y2[v] = fmaf(h1[v] - bn2.mu[v], bn2.sc[v], bn2.be[v]);
        a2[v] = fmaxf(y2[v], 0.f);
      }
      if constexpr (MODE == H_OUT) {
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) part = fmaf(a2[v] * m[v], w4[v], part);
#pragma unroll
        for (int s = 1; s < G::LPR; s <<= 1) part += __shfl_xor(part, s);
        if (lr == 0) a.out[(size_t)b * a.n + n] = part + bias_o;
        continue;
      }
      const float go = a.d_out[(size_t)b * a.n + n];
      float dy2[4], x2h[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        dy2[v] = y2[v] > 0.f ? go * w4[v] * m[v] : 0.f;
        x2h[v] = (h1[v] - bn2.mu[v]) * bn2.is[v];
      }
      if constexpr (MODE == H_BWD2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc0[v] += (double)dy2[v];
          acc1[v] += (double)(dy2[v] * x2h[v]);
          acc2[v] += (double)(go * a2[v] * m[v]);
        }
        if (lr == 0) acc_s += (double)go;
        continue;
      }
      float dh1[4], dy1[4], x1h[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        dh1[v] = bn2.sc[v] * (dy2[v] - m2a[v] - x2h[v] * m2b[v]);
        dy1[v] = y1[v] > 0.f ? dh1[v] * e[v] : 0.f;
        x1h[v] = (z[v] - bn1.mu[v]) * bn1.is[v];
      }
      if constexpr (MODE == H_BWD1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc0[v] += (double)dy1[v];
          acc1[v] += (double)(dy1[v] * x1h[v]);
          const float de = dh1[v] * a1[v];
          if (a.demb_lds) demb_l[n * D + c0 + v] += de;       // (n, column) is private to this thread
          else atomicAdd(a.acc + 6 * D + (size_t)n * D + c0 + v, (double)de);
        }
        continue;
      }
      if constexpr (MODE == H_DZ) {
        float4 o;
        o.x = bn1.sc[0] * (dy1[0] - m1a[0] - x1h[0] * m1b[0]);
        o.y = bn1.sc[1] * (dy1[1] - m1a[1] - x1h[1] * m1b[1]);
        o.z = bn1.sc[2] * (dy1[2] - m1a[2] - x1h[2] * m1b[2]);
        o.w = bn1.sc[3] * (dy1[3] - m1a[3] - x1h[3] * m1b[3]);
        *reinterpret_cast<float4*>(a.d_z + off) = o;
      }
    }
  }

  if constexpr (MODE == H_STAT1 || MODE == H_STAT2) {
    double* dst = a.acc + (MODE == H_STAT1 ? 0 : 2 * D);
    col_reduce<D>(acc0, red, dst, tid, slot, c0);
    col_reduce<D>(acc1, red, dst + D, tid, slot, c0);
  }
  if constexpr (MODE == H_BWD2) {
    col_reduce<D>(acc0, red, a.acc, tid, slot, c0);
    col_reduce<D>(acc1, red, a.acc + D, tid, slot, c0);
    col_reduce<D>(acc2, red, a.acc + 4 * D, tid, slot, c0);
    __syncthreads();
    red[tid] = acc_s;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int q = 0; q < 256; ++q) s += red[q];
      atomicAdd(a.acc + 5 * D, s);
    }
  }
  if constexpr (MODE == H_BWD1) {
    col_reduce<D>(acc0, red, a.acc + 2 * D, tid, slot, c0);
    col_reduce<D>(acc1, red, a.acc + 3 * D, tid, slot, c0);
    if (a.demb_lds) {
      __syncthreads();
      for (int t = tid; t < a.n * D; t += 256) atomicAdd(a.acc + 6 * D + t, (double)demb_l[t]);
    }
  }
}

// running_mean / running_var / num_batches_tracked of both BatchNorms (torch: momentum update with
// the UNBIASED batch variance)
__global__ void gdn_head_running_kernel(const double* __restrict__ fstats, double rows, int d, float mom1,
                                        float* rm1, float* rv1, long long* nbt1, float mom2, float* rm2,
                                        float* rv2, long long* nbt2) {
  const int t = threadIdx.x;
  if (t < d) {
    for (int which = 0; which < 2; ++which) {
      float* rm = which ? rm2 : rm1;
      float* rv = which ? rv2 : rv1;
      const float mom = which ? mom2 : mom1;
      if (!rm || !rv) continue;
      const double* s = fstats + which * 2 * d;
      const double m = s[t] / rows;
      double var = s[d + t] / rows - m * m;
      if (var < 0.0) var = 0.0;
      const double unbiased = var * rows / (rows - 1.0);
      rm[t] = (1.f - mom) * rm[t] + mom * (float)m;
      rv[t] = (1.f - mom) * rv[t] + mom * (float)unbiased;
    }
  }
  if (t == 0) {
    if (nbt1) *nbt1 += 1;
    if (nbt2) *nbt2 += 1;
  }
}

__global__ void gdn_head_finish_kernel(const double* __restrict__ ws, int n, int d, float* d_bn1_w,
                                       float* d_bn1_b, float* d_bn2_w, float* d_bn2_b, float* d_lin_w,
                                       float* d_lin_b, float* d_emb) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < d) {
    d_bn2_b[t] = (float)ws[t];
    d_bn2_w[t] = (float)ws[d + t];
    d_bn1_b[t] = (float)ws[2 * d + t];
    d_bn1_w[t] = (float)ws[3 * d + t];
    d_lin_w[t] = (float)ws[4 * d + t];
    if (t == 0) d_lin_b[0] = (float)ws[5 * d];
  }
  if (t < n * d) d_emb[t] = (float)ws[6 * d + t];
}

int head_grid(int batch) {
  const int cap = 4 * gdn_cu_count();
  return batch < cap ? batch : cap;
}

template <int D, int MODE>
void launch_pass(const HeadArgs& a, hipStream_t st) {
  size_t lds = 1024 * sizeof(double);
  if (MODE == H_BWD1 && a.demb_lds) lds += (size_t)a.n * D * sizeof(float);
  if (lds > 64 * 1024) {
    static bool raised = false;   // per instantiation: allow more than the default 64 KB of dynamic LDS
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gdn_head_train_kernel<D, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        (void)hipGetLastError();
      raised = true;
    }
  }
  hipLaunchKernelGGL((gdn_head_train_kernel<D, MODE>), dim3(head_grid(a.batch)), dim3(256), lds, st, a);
}

bool head_shape_ok(int batch, int n, int d) {
  return batch > 0 && n > 0 && n <= 4096 && (long long)batch * n >= 2;
}

}  // namespace

extern "C" long long gdn_head_train_workspace_bytes(int n, int d) {
  if (n <= 0 || d <= 0) return 0;
  return (long long)(6LL * d + (long long)n * d) * (long long)sizeof(double);
}

extern "C" int gdn_head_train_fwd(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                  const float* bn2_w, const float* bn2_b, const float* lin_w,
                                  const float* lin_b, const float* mask, int batch, int n, int d, float eps1,
                                  float eps2, float momentum1, float momentum2, float* running_mean1,
                                  float* running_var1, long long* batches1, float* running_mean2,
                                  float* running_var2, long long* batches2, double* stats, float* out,
                                  void* stream) {
  if (!z || !emb || !bn1_w || !bn1_b || !bn2_w || !bn2_b || !lin_w || !lin_b || !stats || !out)
    return GDN_ERR_ARG;
  if (!head_shape_ok(batch, n, d)) return GDN_ERR_ARG;   // torch: "Expected more than 1 value per channel"
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  HeadArgs a = {};
  a.z = z; a.emb = emb; a.g1 = bn1_w; a.b1 = bn1_b; a.g2 = bn2_w; a.b2 = bn2_b; a.w = lin_w; a.bo = lin_b;
  a.mask = mask; a.fstats = stats; a.acc = stats; a.out = out; a.batch = batch; a.n = n;
  a.eps1 = eps1; a.eps2 = eps2;
  if (hipMemsetAsync(stats, 0, 4 * (size_t)d * sizeof(double), st) != hipSuccess) return GDN_ERR_LAUNCH;
#define GDN_HEAD_F(DD)                 \
  case DD:                             \
    launch_pass<DD, H_STAT1>(a, st);   \
    launch_pass<DD, H_STAT2>(a, st);   \
    launch_pass<DD, H_OUT>(a, st);     \
    break;
  switch (d) {
    GDN_HEAD_F(16)
    GDN_HEAD_F(32)
    GDN_HEAD_F(64)
    GDN_HEAD_F(128)
  }
#undef GDN_HEAD_F
  if ((running_mean1 && running_var1) || (running_mean2 && running_var2) || batches1 || batches2)
    hipLaunchKernelGGL(gdn_head_running_kernel, dim3(1), dim3(128), 0, st, stats, (double)batch * (double)n, d,
                       momentum1, running_mean1, running_var1, batches1, momentum2, running_mean2,
                       running_var2, batches2);
  return gdn_launch_status();
}

extern "C" int gdn_head_train_bwd(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                                  const float* bn1_b, const float* bn2_w, const float* bn2_b,
                                  const float* lin_w, const float* mask, const double* stats, int batch,
                                  int n, int d, float eps1, float eps2, double* workspace, float* d_z,
                                  float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                                  float* d_bn2_b, float* d_lin_w, float* d_lin_b, void* stream) {
  if (!d_out || !z || !emb || !bn1_w || !bn1_b || !bn2_w || !bn2_b || !lin_w || !stats || !workspace ||
      !d_z || !d_emb || !d_bn1_w || !d_bn1_b || !d_bn2_w || !d_bn2_b || !d_lin_w || !d_lin_b)
    return GDN_ERR_ARG;
  if (!head_shape_ok(batch, n, d)) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  HeadArgs a = {};
  a.z = z; a.emb = emb; a.g1 = bn1_w; a.b1 = bn1_b; a.g2 = bn2_w; a.b2 = bn2_b; a.w = lin_w; a.bo = nullptr;
  a.mask = mask; a.d_out = d_out; a.fstats = stats; a.acc = workspace; a.d_z = d_z; a.batch = batch; a.n = n;
  a.eps1 = eps1; a.eps2 = eps2;
  a.demb_lds = ((size_t)n * d * sizeof(float) + 1024 * sizeof(double)) <= 160 * 1024 ? 1 : 0;
  if (hipMemsetAsync(workspace, 0, (size_t)gdn_head_train_workspace_bytes(n, d), st) != hipSuccess)
    return GDN_ERR_LAUNCH;
#define GDN_HEAD_B(DD)                \
  case DD:                            \
    launch_pass<DD, H_BWD2>(a, st);   \
    launch_pass<DD, H_BWD1>(a, st);   \
    launch_pass<DD, H_DZ>(a, st);     \
    break;
  switch (d) {
    GDN_HEAD_B(16)
    GDN_HEAD_B(32)
    GDN_HEAD_B(64)
    GDN_HEAD_B(128)
  }
#undef GDN_HEAD_B
  const int total = n * d > d ? n * d : d;
  hipLaunchKernelGGL(gdn_head_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, n, d,
                     d_bn1_w, d_bn1_b, d_bn2_w, d_bn2_b, d_lin_w, d_lin_b, d_emb);
  return gdn_launch_status();
}
