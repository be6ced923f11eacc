// Backward of the GraphLayer hot path (training, reference train.py:72 `loss.backward()`
// restricted to models/graph_layer.py:53-117).  Same decomposition as the forward: one
// workgroup per window, 16-lane DPP rows own one target at a time, the (row-offset, value)
// pair of each neighbour rotates through the row.
//
//   z_i = sum_p alpha_ip * xlin[j_p] + bias
//   d_alpha_ip = d_z_i . xlin[j_p]                       (partial dots ride the rotation)
//   d_e_ip     = alpha_ip * (d_alpha_ip - sum_q alpha_iq d_alpha_iq)      (softmax)
//   d_pi_ip    = d_e_ip * (pi_ip > 0 ? 1 : 0.2)                            (LeakyReLU)
//   d_s_i[i]  += sum_p d_pi_ip ;  d_s_j[j_p] += d_pi_ip ;  d_xlin[j_p] += alpha_ip * d_z_i
// Scatter targets (d_xlin, d_s_j) are accumulated in LDS with ds_add_f32 — sources and
// targets of a window live in the same workgroup — and written out once per window.
#include "gdn_common.hpp"

namespace {

template <int V>
struct PackB {
  float v[V];
};
template <int V>
__device__ __forceinline__ PackB<V> ldp(const float* p) {
  PackB<V> r;
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}

template <int D>
struct GeoB {
  static constexpr int DS = D < 64 ? D : 64;
  static constexpr int VEC = DS / 16;
  static constexpr int NS = D / DS;
};

struct BwdPlan {
  int n, d, k, pitch, batch;
  int off_xl, off_dxl, off_sj, off_dsj, off_dbias, off_deg, off_dal, off_nbr;  // float offsets
  int lds_bytes;
};

template <int D>
__global__ __launch_bounds__(256) void gdn_attn_bwd_kernel(
    const BwdPlan pl, const float* __restrict__ d_z, const float* __restrict__ xlin,
    const float* __restrict__ alpha, const float* __restrict__ s_i, const float* __restrict__ s_j,
    const uint16_t* __restrict__ nbr_g, const int32_t* __restrict__ deg_g, float* __restrict__ d_xlin,
    float* __restrict__ d_si, float* __restrict__ d_sj, float* __restrict__ d_bias) {
  using G = GeoB<D>;
  extern __shared__ float4 smem_b4[];
  float* smem = reinterpret_cast<float*>(smem_b4);
  float* xl = smem + pl.off_xl;
  float* dxl = smem + pl.off_dxl;
  float* sj = smem + pl.off_sj;
  float* dsj = smem + pl.off_dsj;
  float* dbias = smem + pl.off_dbias;
  uint16_t* degs = reinterpret_cast<uint16_t*>(smem + pl.off_deg);
  uint16_t* nbr = reinterpret_cast<uint16_t*>(smem + pl.off_nbr);
  const int tid = threadIdx.x, nth = blockDim.x;
  const int grp = tid >> 4, l16 = tid & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (nth >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  float* dal = smem + pl.off_dal + grp * pl.pitch;  // this row's d_alpha scratch

  for (int t = tid; t < pl.n; t += nth) degs[t] = (uint16_t)deg_g[t];
  {
    const uint4* src = reinterpret_cast<const uint4*>(nbr_g);
    uint4* dst = reinterpret_cast<uint4*>(nbr);
    const int nvec = pl.n * pl.pitch / 8;
    for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
  }
  for (int t = tid; t < D; t += nth) {
    dbias[t] = 0.f;
    xl[pl.n * D + t] = 0.f;   // sentinel row: read by padding slots, weight 0
  }
  if (tid == 0) sj[pl.n] = 0.f;
  PackB<G::VEC> bias_acc;
#pragma unroll
  for (int v = 0; v < G::VEC; ++v) bias_acc.v[v] = 0.f;

  for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
    const size_t row0 = (size_t)b * pl.n;
    {
      const float4* src = reinterpret_cast<const float4*>(xlin + row0 * D);
      float4* dst = reinterpret_cast<float4*>(xl);
      float4* zdst = reinterpret_cast<float4*>(dxl);
      const int nvec = pl.n * D / 4;
      for (int t = tid; t < nvec; t += nth) {
        dst[t] = src[t];
        zdst[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int t = tid; t < pl.n; t += nth) {
        sj[t] = s_j[row0 + t];
        dsj[t] = 0.f;
      }
    }
    __syncthreads();

    for (int i = slot; i < pl.n; i += tpp) {
      const int degi = degs[i];
      const int rounds = (degi + 15) >> 4;
      const uint16_t* nrow = nbr + (size_t)i * pl.pitch;
      const float* arow = alpha + (row0 + i) * pl.pitch;
      const PackB<G::VEC> g = ldp<G::VEC>(d_z + (row0 + i) * D + d0);
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) bias_acc.v[v] += g.v[v];
      const float sti = s_i[row0 + i];

      // pass A: d_alpha of every neighbour (partial dots ride the rotation), and sum alpha*d_alpha
      float dot = 0.f;
      for (int r = 0; r < rounds; ++r) {
        const int p = r * 16 + l16;
        int jb = nrow[p] * (D * 4);
        float tsum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const PackB<G::VEC> src =
              ldp<G::VEC>(reinterpret_cast<const float*>(reinterpret_cast<const char*>(xl + d0) + jb));
#pragma unroll
          for (int v = 0; v < G::VEC; ++v) tsum = fmaf(g.v[v], src.v[v], tsum);
          tsum = dpp_f<GDN_DPP_ROR1>(tsum);
          jb = dpp_i<GDN_DPP_ROR1>(jb);
        }
        // after 16 steps tsum is back on its home lane holding the full dot over this slice
        if constexpr (G::NS == 2) tsum += __shfl_xor(tsum, 16);
        const float al = p < degi ? arow[p] : 0.f;
        dot = fmaf(al, tsum, dot);
        dal[p] = tsum;  // per-lane scratch: read back only by this lane in pass B
      }
      dot = row16_sum(dot);
      // pass B: logits' gradients and the two scatters
      float dsi = 0.f;
      for (int r = 0; r < rounds; ++r) {
        const int p = r * 16 + l16;
        const int j = nrow[p];
        const float al = p < degi ? arow[p] : 0.f;
        const float dalp = dal[p];
        const float de = al * (dalp - dot);
        const float pi = sti + sj[j];
        const float dpi = de * (pi > 0.f ? 1.f : GDN_NEG_SLOPE);
        if (p < degi && slice == 0) {
          dsi += dpi;
          atomicAdd(&dsj[j], dpi);  // ds_add_f32
        }
        float a_rot = al;
        int jb = j * (D * 4);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(dxl + d0) + jb);
          if (a_rot != 0.f) {
#pragma unroll
            for (int v = 0; v < G::VEC; ++v) atomicAdd(dst + v, a_rot * g.v[v]);
          }
          a_rot = dpp_f<GDN_DPP_ROR1>(a_rot);
          jb = dpp_i<GDN_DPP_ROR1>(jb);
        }
      }
      dsi = row16_sum(dsi);
      if (l16 == 0 && slice == 0) d_si[row0 + i] = dsi;
    }
    __syncthreads();
    {
      const float4* src = reinterpret_cast<const float4*>(dxl);
      float4* dst = reinterpret_cast<float4*>(d_xlin + row0 * D);
      const int nvec = pl.n * D / 4;
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
      for (int t = tid; t < pl.n; t += nth) d_sj[row0 + t] = dsj[t];
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < G::VEC; ++v) atomicAdd(&dbias[d0 + v], bias_acc.v[v]);
  __syncthreads();
  for (int t = tid; t < D; t += nth) atomicAdd(&d_bias[t], dbias[t]);
}

// d_lin_w[d,w] += sum_rows d_xlin[row,d] x[row,w];  d_a[2,64] += sum_rows d_s[row] x[row,:];
// d_c[2,n] += sum_b d_s[b*n + s].
template <int D, int WCH>
__global__ __launch_bounds__(256) void gdn_project_bwd_kernel(
    int batch, int n, int w, int wp, const float* __restrict__ x, const float* __restrict__ d_xlin,
    const float* __restrict__ d_si, const float* __restrict__ d_sj, float* __restrict__ d_lin_w,
    float* __restrict__ d_a, float* __restrict__ d_c) {
  using G = GeoB<D>;
  extern __shared__ float4 smem_b4[];
  float* smem = reinterpret_cast<float*>(smem_b4);
  float* xs = smem;                 // [n][wp]
  float* dw = xs + (size_t)n * wp;  // [D][wp] workgroup partial of d_lin_w
  float* da = dw + (size_t)D * wp;  // [2][64]
  float* dc = da + 128;             // [2][n]
  const int tid = threadIdx.x, nth = blockDim.x;
  const int grp = tid >> 4, l16 = tid & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (nth >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  const int nch = wp / WCH;

  for (int t = tid; t < D * wp + 128 + 2 * n; t += nth) dw[t] = 0.f;
  __syncthreads();

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const size_t row0 = (size_t)b * n;
    const float* xg = x + row0 * w;
    for (int t = tid; t < n * wp; t += nth) {
      const int r = t / wp, c = t - r * wp;
      xs[t] = c < w ? xg[(size_t)r * w + c] : 0.f;
    }
    for (int t = tid; t < n; t += nth) {
      dc[t] += d_si[row0 + t];
      dc[n + t] += d_sj[row0 + t];
    }
    __syncthreads();
    for (int wc = 0; wc < nch; ++wc) {
      float acc[G::VEC][WCH];
#pragma unroll
      for (int v = 0; v < G::VEC; ++v)
#pragma unroll
        for (int c = 0; c < WCH; ++c) acc[v][c] = 0.f;
      float ai = 0.f, aj = 0.f;
      const bool ahas = l16 < WCH;
      for (int row = slot; row < n; row += tpp) {
        const float* xrow = xs + (size_t)row * wp + wc * WCH;
        float xr[WCH];
#pragma unroll
        for (int c = 0; c < WCH; c += 4) {
          const float4 t = *reinterpret_cast<const float4*>(xrow + c);
          xr[c] = t.x; xr[c + 1] = t.y; xr[c + 2] = t.z; xr[c + 3] = t.w;
        }
        const PackB<G::VEC> g = ldp<G::VEC>(d_xlin + (row0 + row) * D + d0);
#pragma unroll
        for (int v = 0; v < G::VEC; ++v)
#pragma unroll
          for (int c = 0; c < WCH; ++c) acc[v][c] = fmaf(g.v[v], xr[c], acc[v][c]);
        if (slice == 0 && ahas) {
          const float xv = xrow[l16];
          ai = fmaf(d_si[row0 + row], xv, ai);
          aj = fmaf(d_sj[row0 + row], xv, aj);
        }
      }
#pragma unroll
      for (int v = 0; v < G::VEC; ++v)
#pragma unroll
        for (int c = 0; c < WCH; ++c) atomicAdd(&dw[(size_t)(d0 + v) * wp + wc * WCH + c], acc[v][c]);
      if (slice == 0 && ahas) {
        atomicAdd(&da[wc * WCH + l16], ai);
        atomicAdd(&da[64 + wc * WCH + l16], aj);
      }
    }
    __syncthreads();
  }
  // one flush per workgroup
  for (int t = tid; t < D * wp; t += nth) {
    const int r = t / wp, c = t - r * wp;
    if (c < w) atomicAdd(&d_lin_w[(size_t)r * w + c], dw[t]);
  }
  for (int t = tid; t < 128; t += nth) atomicAdd(&d_a[t], da[t]);
  for (int t = tid; t < 2 * n; t += nth) atomicAdd(&d_c[t], dc[t]);
}

template <typename K>
int occupancy_grid(K kern, int threads, int lds, int batch) {
  int nb = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess)
    (void)hipGetLastError();
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb <= 0) {
    (void)hipGetLastError();
    nb = 1;
  }
  return min(batch, gdn_cu_count() * nb);
}

}  // namespace

extern "C" int gdn_attn_aggregate_bwd(const float* d_z, const float* xlin, const float* alpha,
                                      const float* s_i, const float* s_j, const uint16_t* nbr,
                                      const int32_t* deg, int batch, int n, int d, int k, float* d_xlin,
                                      float* d_si, float* d_sj, float* d_bias, void* stream) {
  if (!d_z || !xlin || !alpha || !s_i || !s_j || !nbr || !deg || !d_xlin || !d_si || !d_sj || !d_bias ||
      batch <= 0 || n <= 0 || k <= 0)
    return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (k > n || n > 4096 || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
  BwdPlan pl;
  pl.n = n; pl.d = d; pl.k = k; pl.batch = batch; pl.pitch = gdn_nbr_pitch(k);
  const int npad = (n + 1 + 3) & ~3;   // +1: the sentinel index n used as list padding
  int off = 0;
  pl.off_xl = off; off += (n + 1) * d;
  pl.off_dxl = off; off += (n + 1) * d;
  pl.off_sj = off; off += npad;
  pl.off_dsj = off; off += npad;
  pl.off_dbias = off; off += d;
  pl.off_deg = off; off += (npad / 2 + 3) & ~3;
  pl.off_dal = off; off += 16 * pl.pitch;
  pl.off_nbr = off; off += n * pl.pitch / 2;
  pl.lds_bytes = off * 4;
  if (pl.lds_bytes > 160 * 1024) return GDN_ERR_UNSUPPORTED;  // TODO: global-atomic variant for big tiles
  hipStream_t st = (hipStream_t)stream;
#define GDN_BWD(DD)                                                                                  \
  case DD: {                                                                                         \
    const int grid = occupancy_grid(gdn_attn_bwd_kernel<DD>, 256, pl.lds_bytes, batch);              \
    hipLaunchKernelGGL(gdn_attn_bwd_kernel<DD>, dim3(grid), dim3(256), pl.lds_bytes, st, pl, d_z, xlin, \
                       alpha, s_i, s_j, nbr, deg, d_xlin, d_si, d_sj, d_bias);                       \
  } break;
  switch (d) {
    GDN_BWD(16)
    GDN_BWD(32)
    GDN_BWD(64)
    GDN_BWD(128)
  }
#undef GDN_BWD
  return gdn_launch_status();
}

extern "C" int gdn_project_bwd(const float* x, const float* d_xlin, const float* d_si, const float* d_sj,
                               int batch, int n, int w, int d, float* d_lin_w, float* d_a, float* d_c,
                               void* stream) {
  if (!x || !d_xlin || !d_si || !d_sj || !d_lin_w || !d_a || !d_c || batch <= 0 || n <= 0 || w <= 0)
    return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (w > GDN_MAX_W || n > 4096) return GDN_ERR_UNSUPPORTED;
  const int wp = w <= 8 ? 8 : ((w + 15) & ~15);
  const int lds = (n * wp + d * wp + 128 + 2 * n) * 4;
  if (lds > 160 * 1024) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define GDN_PB(DD, WW)                                                                              \
  {                                                                                                 \
    const int grid = occupancy_grid(gdn_project_bwd_kernel<DD, WW>, 256, lds, batch);               \
    hipLaunchKernelGGL((gdn_project_bwd_kernel<DD, WW>), dim3(grid), dim3(256), lds, st, batch, n, w, wp, \
                       x, d_xlin, d_si, d_sj, d_lin_w, d_a, d_c);                                   \
  }
#define GDN_PBD(DD)          \
  case DD:                   \
    if (wp == 8) GDN_PB(DD, 8) else GDN_PB(DD, 16) break;
  switch (d) {
    GDN_PBD(16)
    GDN_PBD(32)
    GDN_PBD(64)
    GDN_PBD(128)
  }
#undef GDN_PBD
#undef GDN_PB
  return gdn_launch_status();
}
