// Backward of the GraphLayer hot path (training, reference train.py:72 `loss.backward()`
// restricted to models/graph_layer.py:53-117).  Same decomposition as the forward: one
// workgroup per window, 16-lane DPP rows own one sensor at a time, (row-offset, weight) pairs
// rotate through the row.  No scatter: both directions are GATHERS.
//
//   z_i = sum_p alpha_ip * xlin[j_p] + bias
//   pass 1 (per TARGET i, xlin tile in LDS):
//     d_alpha_ip = d_z_i . xlin[j_p]          partial dots of the 16 lanes are rotated home
//     d_e_ip     = alpha_ip * (d_alpha_ip - sum_q alpha_iq d_alpha_iq)          (softmax)
//     d_pi_ip    = d_e_ip * (pi_ip > 0 ? 1 : 0.2)                               (LeakyReLU)
//     d_s_i[i]   = sum_p d_pi_ip ;  alpha and d_pi are kept in two [n, pitch] LDS tables
//   pass 2 (per SOURCE j, d_z tile in LDS, REVERSE lists: the (target, slot) pairs that name j):
//     d_xlin[j]  = sum_(i,p) alpha_ip * d_z_i ;  d_s_j[j] = sum_(i,p) d_pi_ip
// Results are bitwise reproducible (fixed list order); the first version scattered with ds_add_f32
// and spent 650 us per 512 windows where the forward needs 17.
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
template <int V>
__device__ __forceinline__ void stp(float* p, const PackB<V>& r) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else if constexpr (V == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  else *p = r.v[0];
}

template <int D>
struct GeoB {
  static constexpr int DS = D < 64 ? D : 64;
  static constexpr int VEC = DS / 16;
  static constexpr int NS = D / DS;
};

template <int ROT>
__device__ __forceinline__ int rorb_i(int v) {
  if constexpr (ROT == 0) return v;
  else return dpp_i<0x120 + ROT>(v);
}
template <int ROT>
__device__ __forceinline__ float rorb_f(float v) {
  if constexpr (ROT == 0) return v;
  else return dpp_f<0x120 + ROT>(v);
}

// tsum (home lane q) = sum over the 16 lanes of g . tile[row of neighbour q][lane's columns]:
// at step S lane l holds neighbour (l+S)%16's row offset; its partial dot is rotated by 16-S lanes
// back to the neighbour's home lane.  Every step is independent of the others.
template <int D, int S = 0>
__device__ __forceinline__ void dot_steps(const char* tile_lane, int jb, const PackB<GeoB<D>::VEC>& g,
                                          float& tsum) {
  if constexpr (S < 16) {
    const PackB<GeoB<D>::VEC> src =
        ldp<GeoB<D>::VEC>(reinterpret_cast<const float*>(tile_lane + rorb_i<S>(jb)));
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < GeoB<D>::VEC; ++v) part = fmaf(g.v[v], src.v[v], part);
    tsum += rorb_f<(16 - S) & 15>(part);
    if constexpr (S == 7) __builtin_amdgcn_sched_barrier(0);
    dot_steps<D, S + 1>(tile_lane, jb, g, tsum);
  }
}

// acc += a_q * tile[row q] for the 16 (weight, row) pairs held by the lanes of this row
template <int D, int S = 0>
__device__ __forceinline__ void axpy_steps(const char* tile_lane, float a, int rb, PackB<GeoB<D>::VEC>& acc) {
  if constexpr (S < 16) {
    const PackB<GeoB<D>::VEC> src =
        ldp<GeoB<D>::VEC>(reinterpret_cast<const float*>(tile_lane + rorb_i<S>(rb)));
    const float a_s = rorb_f<S>(a);
#pragma unroll
    for (int v = 0; v < GeoB<D>::VEC; ++v) acc.v[v] = fmaf(a_s, src.v[v], acc.v[v]);
    if constexpr (S == 7) __builtin_amdgcn_sched_barrier(0);
    axpy_steps<D, S + 1>(tile_lane, a, rb, acc);
  }
}

struct BwdPlan {
  int n, d, k, pitch, rpitch, batch;
  int off_tile, off_sj, off_al, off_dpi, off_dbias;  // float offsets
  int lds_bytes;
};

template <int D>
__global__ __launch_bounds__(256) void gdn_attn_bwd_kernel(
    const BwdPlan pl, const float* __restrict__ d_z, const float* __restrict__ xlin,
    const float* __restrict__ alpha, const float* __restrict__ s_i, const float* __restrict__ s_j,
    const uint16_t* __restrict__ nbr, const uint32_t* __restrict__ rent, const int32_t* __restrict__ rlen,
    float* __restrict__ d_xlin, float* __restrict__ d_si, float* __restrict__ d_sj,
    float* __restrict__ d_bias) {
  using G = GeoB<D>;
  extern __shared__ float4 smem_b4[];
  float* smem = reinterpret_cast<float*>(smem_b4);
  float* tile = smem + pl.off_tile;    // xlin (pass 1) then d_z (pass 2); row n stays 0
  float* sj = smem + pl.off_sj;
  float* al_t = smem + pl.off_al;      // alpha  [n, pitch]
  float* dpi_t = smem + pl.off_dpi;    // d_pi   [n, pitch]
  float* dbias = smem + pl.off_dbias;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int grp = tid >> 4, l16 = tid & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (nth >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  const char* tile_lane = reinterpret_cast<const char*>(tile + d0);
  const int rounds = pl.pitch >> 4;

  for (int t = tid; t < D; t += nth) {
    dbias[t] = 0.f;
    tile[pl.n * D + t] = 0.f;   // sentinel row: read by padding slots, weight 0
  }
  if (tid == 0) sj[pl.n] = 0.f;
  PackB<G::VEC> bias_acc;
#pragma unroll
  for (int v = 0; v < G::VEC; ++v) bias_acc.v[v] = 0.f;
  const int nvec = pl.n * D / 4;

  for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
    const size_t row0 = (size_t)b * pl.n;
    {
      const float4* src = reinterpret_cast<const float4*>(xlin + row0 * D);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
      for (int t = tid; t < pl.n; t += nth) sj[t] = s_j[row0 + t];
    }
    __syncthreads();

    // ---- pass 1: per target
    for (int i = slot; i < pl.n; i += tpp) {
      const uint16_t* nrow = nbr + (size_t)i * pl.pitch;
      const float* arow = alpha + (row0 + i) * pl.pitch;
      const PackB<G::VEC> g = ldp<G::VEC>(d_z + (row0 + i) * D + d0);
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) bias_acc.v[v] += g.v[v];
      const float sti = s_i[row0 + i];
      float dot = 0.f;
      for (int r = 0; r < rounds; ++r) {
        const int p = r * 16 + l16;
        const int j = nrow[p];            // padding = sentinel n (zero row, alpha 0)
        float tsum = 0.f;
        dot_steps<D>(tile_lane, j * (D * 4), g, tsum);
        if constexpr (G::NS == 2) tsum += __shfl_xor(tsum, 16);
        const float al = arow[p];
        dot = fmaf(al, tsum, dot);
        if (slice == 0) {
          al_t[i * pl.pitch + p] = al;
          dpi_t[i * pl.pitch + p] = tsum;   // d_alpha for now; finished below
        }
      }
      dot = row16_sum(dot);
      float dsi = 0.f;
      if (slice == 0) {
        for (int r = 0; r < rounds; ++r) {
          const int p = r * 16 + l16;
          const int j = nrow[p];
          const float al = al_t[i * pl.pitch + p];      // written by this lane above
          const float de = al * (dpi_t[i * pl.pitch + p] - dot);
          const float pi = sti + sj[j];
          const float dpi = de * (pi > 0.f ? 1.f : GDN_NEG_SLOPE);
          dpi_t[i * pl.pitch + p] = dpi;
          dsi += dpi;
        }
      }
      dsi = row16_sum(dsi);
      if (l16 == 0 && slice == 0) d_si[row0 + i] = dsi;
    }
    __syncthreads();
    {   // the tile now holds d_z of this window (row n stays 0)
      const float4* src = reinterpret_cast<const float4*>(d_z + row0 * D);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
    }
    __syncthreads();

    // ---- pass 2: per source, over its reverse list
    for (int j = slot; j < pl.n; j += tpp) {
      const int len = rlen[j];
      const uint32_t* rrow = rent + (size_t)j * pl.rpitch;
      PackB<G::VEC> acc;
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) acc.v[v] = 0.f;
      float dsj = 0.f;
      for (int r0 = 0; r0 < len; r0 += 16) {
        const int e = r0 + l16;
        const bool valid = e < len;
        const uint32_t ent = rrow[valid ? e : 0];
        const int i = ent >> 16, p = ent & 0xffff;
        const float a = valid ? al_t[i * pl.pitch + p] : 0.f;
        dsj += valid ? dpi_t[i * pl.pitch + p] : 0.f;
        axpy_steps<D>(tile_lane, a, (valid ? i : pl.n) * (D * 4), acc);
      }
      stp<G::VEC>(d_xlin + (row0 + j) * D + d0, acc);
      dsj = row16_sum(dsj);
      if (l16 == 0 && slice == 0) d_sj[row0 + j] = dsj;
    }
    __syncthreads();   // tables and tile are rewritten by the next window
  }
#pragma unroll
  for (int v = 0; v < G::VEC; ++v) atomicAdd(&dbias[d0 + v], bias_acc.v[v]);
  __syncthreads();
  for (int t = tid; t < D; t += nth) atomicAdd(&d_bias[t], dbias[t]);
}

// Reverse lists: for source j, the (target i, slot p) pairs with nbr[i][p] == j, in ascending i
// (a source appears at most once per target).  One block per source; rent[j, rpitch] u32 = i<<16 | p.
__global__ __launch_bounds__(256) void gdn_graph_reverse_kernel(const uint16_t* __restrict__ nbr,
                                                                const int32_t* __restrict__ deg, int n,
                                                                int pitch, int rpitch,
                                                                uint32_t* __restrict__ rent,
                                                                int32_t* __restrict__ rlen) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int j = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    int found = -1;
    if (i < n) {
      const int dg = deg[i];
      for (int p = 0; p < dg; ++p)
        if (nbr[(size_t)i * pitch + p] == j) found = p;
    }
    const unsigned long long m = __ballot(found >= 0);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int q = 0; q < wv; ++q) off += wave_cnt[q];
    if (found >= 0) rent[(size_t)j * rpitch + off + before] = ((uint32_t)i << 16) | (uint32_t)found;
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) rlen[j] = base_s;
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

  // accumulators live in registers across ALL windows of this workgroup and are flushed once per
  // W-chunk (flushing per window cost 16k ds_add_f32 per window)
  for (int wc = 0; wc < nch; ++wc) {
    float acc[G::VEC][WCH];
#pragma unroll
    for (int v = 0; v < G::VEC; ++v)
#pragma unroll
      for (int c = 0; c < WCH; ++c) acc[v][c] = 0.f;
    float ai = 0.f, aj = 0.f;
    const bool ahas = l16 < WCH;
    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
      const size_t row0 = (size_t)b * n;
      const float* xg = x + row0 * w;
      for (int t = tid; t < n * wp; t += nth) {
        const int r = t / wp, c = t - r * wp;
        xs[t] = c < w ? xg[(size_t)r * w + c] : 0.f;
      }
      if (wc == 0) {
        for (int t = tid; t < n; t += nth) {
          dc[t] += d_si[row0 + t];
          dc[n + t] += d_sj[row0 + t];
        }
      }
      __syncthreads();
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
      __syncthreads();   // the x tile is restaged for the next window
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
  // one flush per workgroup
  for (int t = tid; t < D * wp; t += nth) {
    const int r = t / wp, c = t - r * wp;
    if (c < w) atomicAdd(&d_lin_w[(size_t)r * w + c], dw[t]);
  }
  for (int t = tid; t < 128; t += nth) atomicAdd(&d_a[t], da[t]);
  for (int t = tid; t < 2 * n; t += nth) atomicAdd(&d_c[t], dc[t]);
}

// Chain rule through the folded constants a = lin^T att (node_terms) and c = emb . att_em:
// d_lin_w += att_i (x) d_a[0] + att_j (x) d_a[1];  d_att = lin_w d_a;  d_att_em = emb^T d_c;
// d_emb = d_c[0] (x) att_em_i + d_c[1] (x) att_em_j.  Tiny ([d,w], [n,d]); one launch instead of a dozen.
__global__ __launch_bounds__(256) void gdn_terms_bwd_kernel(
    const float* __restrict__ lin_w, const float* __restrict__ att_i, const float* __restrict__ att_j,
    const float* __restrict__ att_em_i, const float* __restrict__ att_em_j, const float* __restrict__ emb,
    const float* __restrict__ d_a, const float* __restrict__ d_c, int n, int d, int w,
    float* __restrict__ d_lin_w, float* __restrict__ d_att_i, float* __restrict__ d_att_j,
    float* __restrict__ d_att_em_i, float* __restrict__ d_att_em_j, float* __restrict__ d_emb) {
  const int tid = threadIdx.x;
  for (int t = blockIdx.x * 256 + tid; t < n * d; t += gridDim.x * 256) {
    const int s = t / d, c = t - s * d;
    d_emb[t] = fmaf(d_c[s], att_em_i[c], d_c[n + s] * att_em_j[c]);
  }
  if (blockIdx.x != 0) return;
  __shared__ float part[4][256];
  for (int t = tid; t < d * w; t += 256) {
    const int c = t / w, q = t - c * w;
    d_lin_w[t] += fmaf(att_i[c], d_a[q], att_j[c] * d_a[GDN_A_PITCH + q]);
  }
  // column c is handled by the 256/d threads tid = c, c+d, ...: strided partial sums, then an LDS reduce
  const int c = tid % d, g = tid / d, groups = 256 / d;
  float si = 0.f, sj = 0.f, ei = 0.f, ej = 0.f;
  for (int q = g; q < w; q += groups) {
    const float lw = lin_w[c * w + q];
    si = fmaf(lw, d_a[q], si);
    sj = fmaf(lw, d_a[GDN_A_PITCH + q], sj);
  }
  for (int s = g; s < n; s += groups) {
    const float ev = emb[(size_t)s * d + c];
    ei = fmaf(ev, d_c[s], ei);
    ej = fmaf(ev, d_c[n + s], ej);
  }
  part[0][tid] = si; part[1][tid] = sj; part[2][tid] = ei; part[3][tid] = ej;
  __syncthreads();
  if (tid < d) {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < groups; ++q)
#pragma unroll
      for (int v = 0; v < 4; ++v) r[v] += part[v][q * d + tid];
    d_att_i[tid] = r[0];
    d_att_j[tid] = r[1];
    d_att_em_i[tid] = r[2];
    d_att_em_j[tid] = r[3];
  }
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

extern "C" int gdn_rev_pitch(int n) { return (n + 15) & ~15; }

extern "C" int gdn_graph_reverse(const uint16_t* nbr, const int32_t* deg, int n, int k, uint32_t* rent,
                                 int32_t* rlen, void* stream) {
  if (!nbr || !deg || !rent || !rlen || n <= 0 || k <= 0) return GDN_ERR_ARG;
  if (k > n || n > 4096 || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gdn_graph_reverse_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, nbr, deg, n,
                     gdn_nbr_pitch(k), gdn_rev_pitch(n), rent, rlen);
  return gdn_launch_status();
}

extern "C" int gdn_attn_aggregate_bwd(const float* d_z, const float* xlin, const float* alpha,
                                      const float* s_i, const float* s_j, const uint16_t* nbr,
                                      const uint32_t* rent, const int32_t* rlen, int batch, int n, int d,
                                      int k, float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                                      void* stream) {
  if (!d_z || !xlin || !alpha || !s_i || !s_j || !nbr || !rent || !rlen || !d_xlin || !d_si || !d_sj ||
      !d_bias || batch <= 0 || n <= 0 || k <= 0)
    return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (k > n || n > 4096 || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
  BwdPlan pl;
  pl.n = n; pl.d = d; pl.k = k; pl.batch = batch; pl.pitch = gdn_nbr_pitch(k); pl.rpitch = gdn_rev_pitch(n);
  const int npad = (n + 1 + 3) & ~3;   // +1: the sentinel index n used as list padding
  int off = 0;
  pl.off_tile = off; off += (n + 1) * d;
  pl.off_sj = off; off += npad;
  pl.off_al = off; off += n * pl.pitch;
  pl.off_dpi = off; off += n * pl.pitch;
  pl.off_dbias = off; off += d;
  pl.lds_bytes = off * 4;
  if (pl.lds_bytes > 160 * 1024) return GDN_ERR_UNSUPPORTED;   // n*d tile + two [n,pitch] tables
  hipStream_t st = (hipStream_t)stream;
#define GDN_BWD(DD)                                                                                  \
  case DD: {                                                                                         \
    const int grid = occupancy_grid(gdn_attn_bwd_kernel<DD>, 256, pl.lds_bytes, batch);              \
    hipLaunchKernelGGL(gdn_attn_bwd_kernel<DD>, dim3(grid), dim3(256), pl.lds_bytes, st, pl, d_z, xlin, \
                       alpha, s_i, s_j, nbr, rent, rlen, d_xlin, d_si, d_sj, d_bias);                \
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

extern "C" int gdn_terms_bwd(const float* lin_w, const float* att_i, const float* att_j,
                             const float* att_em_i, const float* att_em_j, const float* emb,
                             const float* d_a, const float* d_c, int n, int d, int w, float* d_lin_w,
                             float* d_att_i, float* d_att_j, float* d_att_em_i, float* d_att_em_j,
                             float* d_emb, void* stream) {
  if (!lin_w || !att_i || !att_j || !att_em_i || !att_em_j || !emb || !d_a || !d_c || !d_lin_w || !d_att_i ||
      !d_att_j || !d_att_em_i || !d_att_em_j || !d_emb || n <= 0 || d <= 0 || w <= 0)
    return GDN_ERR_ARG;
  if (w > GDN_MAX_W || d > 256 || (256 % d) != 0) return GDN_ERR_UNSUPPORTED;
  int grid = (n * d + 256 * 8 - 1) / (256 * 8);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(gdn_terms_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, lin_w, att_i, att_j,
                     att_em_i, att_em_j, emb, d_a, d_c, n, d, w, d_lin_w, d_att_i, d_att_j, d_att_em_i,
                     d_att_em_j, d_emb);
  return gdn_launch_status();
}
