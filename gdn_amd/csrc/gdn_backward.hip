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
#include <stdlib.h>

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
  int off_tile, off_sj, off_si, off_al, off_dpi, off_dbias, off_nbr, off_red;  // float offsets
  int lds_bytes;
};

// NT threads per workgroup: 512 (8 waves, 4 per SIMD with two workgroups per CU) hides the LDS / DPP
// latency chains of the two gather loops better than 256; BL = targets / sources a lane group
// prefetches for before it starts computing (fewer at 512 threads: 128 VGPRs per lane).
// GLB = the two [n, pitch] tables and the lists do not fit beside the tile (large n: 512 sensors x 80
// slots are 164 KB per table): alpha and the lists are read from global memory (L2 resident: one window's
// worth), d_pi goes through a caller-provided [BN, pitch] workspace (written in pass 1, read in pass 2 by
// other lanes of the SAME workgroup, a barrier in between).  Same arithmetic, same summation order.
// DX = false: d_xlin is left to gdn_dense_attn_bwd_dx (the matrix-core form of pass 2); this kernel then
// only sums d_s_j over the reverse lists in its second pass.
// NSL > 1 (GLB only): a window's [n, D*NSL] tile does not fit LDS (512 sensors x 128 columns are 263 KB): the
// workgroup walks NSL column slices of D one after the other, in both passes.  Pass 1 keeps the partial
// d_alpha of the earlier slices in the d_pi workspace and finishes the softmax on the last slice (each
// element is written and read back by the same lane); the sum of two slices is the sum the unsliced kernel
// forms with its two lane groups, so both give the same bits.
template <int D, int NT, bool GLB, bool DX = true, int NSL = 1>
__global__ __launch_bounds__(NT) void gdn_attn_bwd_kernel(
    const BwdPlan pl, const float* __restrict__ d_z, const float* __restrict__ xlin,
    const float* __restrict__ alpha, const float* __restrict__ s_i, const float* __restrict__ s_j,
    const uint16_t* __restrict__ nbr, const uint32_t* __restrict__ rent, const int32_t* __restrict__ rlen,
    float* __restrict__ d_xlin, float* __restrict__ d_si, float* __restrict__ d_sj,
    float* __restrict__ d_bias, float* __restrict__ dpi_ws, float* __restrict__ bias_ws) {
  using G = GeoB<D>;
  constexpr int BL = 2048 / NT;
  constexpr int DF = D * NSL;          // row stride of the global arrays
  static_assert(NSL == 1 || (GLB && G::NS == 1), "column slices: tables in global memory, 64-column tiles");
  extern __shared__ float4 smem_b4[];
  float* smem = reinterpret_cast<float*>(smem_b4);
  float* tile = smem + pl.off_tile;    // xlin (pass 1) then d_z (pass 2); row n stays 0
  float* sj = smem + pl.off_sj;
  float* si = smem + pl.off_si;
  float* dbias = smem + pl.off_dbias;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int grp = tid >> 4, l16 = tid & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (nth >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  const char* tile_lane = reinterpret_cast<const char*>(tile + d0);
  const int rounds = pl.pitch >> 4;
  const int npl = pl.n * pl.pitch;

  for (int t = tid; t < DF; t += nth) dbias[t] = 0.f;
  for (int t = tid; t < D; t += nth) tile[pl.n * D + t] = 0.f;   // sentinel row: read by padding slots, weight 0
  if (tid == 0) sj[pl.n] = 0.f;
  if constexpr (!GLB)
    for (int t = tid; t < npl / 2; t += nth)   // pitch is a multiple of 16: copy the lists as u32 pairs
      reinterpret_cast<uint32_t*>(smem + pl.off_nbr)[t] = reinterpret_cast<const uint32_t*>(nbr)[t];
  PackB<G::VEC> bias_acc[NSL];
#pragma unroll
  for (int s = 0; s < NSL; ++s)
#pragma unroll
    for (int v = 0; v < G::VEC; ++v) bias_acc[s].v[v] = 0.f;
  const int nvec = pl.n * D / 4;
  // tile <- columns [c0, c0 + D) of the window's rows of `src` (row stride DF)
  auto stage_tile = [&](const float* src, int c0) {
    float4* dst = reinterpret_cast<float4*>(tile);
    if constexpr (NSL == 1) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll 4
      for (int t = tid; t < nvec; t += nth) dst[t] = s4[t];
    } else {
      const float4* s4 = reinterpret_cast<const float4*>(src + c0);
#pragma unroll 4
      for (int t = tid; t < nvec; t += nth) dst[t] = s4[(t / (D / 4)) * (DF / 4) + (t % (D / 4))];
    }
  };

  for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
    const size_t row0 = (size_t)b * pl.n;
    // the three per-window tables: LDS copies, or (GLB) the global arrays themselves
    const float* al_t;            // alpha  [n, pitch]
    float* dpi_t;                 // d_alpha, then d_pi   [n, pitch]
    const uint16_t* nb_l;         // neighbour lists [n, pitch]
    if constexpr (GLB) {
      al_t = alpha + row0 * pl.pitch;
      dpi_t = dpi_ws + row0 * pl.pitch;
      nb_l = nbr;
    } else {
      al_t = smem + pl.off_al;
      dpi_t = smem + pl.off_dpi;
      nb_l = reinterpret_cast<const uint16_t*>(smem + pl.off_nbr);
    }
    {   // bulk staging (many loads in flight): xlin tile, alpha table, s_i, s_j
      stage_tile(xlin + row0 * DF, 0);
      if constexpr (!GLB) {
        const float4* asrc = reinterpret_cast<const float4*>(alpha + row0 * pl.pitch);
        float4* adst = reinterpret_cast<float4*>(smem + pl.off_al);
#pragma unroll 4
        for (int t = tid; t < npl / 4; t += nth) adst[t] = asrc[t];
      }
      for (int t = tid; t < pl.n; t += nth) {
        sj[t] = s_j[row0 + t];
        si[t] = s_i[row0 + t];
      }
    }
    __syncthreads();

    // ---- pass 1: per target, BL targets per lane group at a time (their d_z rows are fetched together)
#pragma unroll
    for (int cs = 0; cs < NSL; ++cs) {
    if (cs > 0) {        // next column slice of xlin
      __syncthreads();
      stage_tile(xlin + row0 * DF, cs * D);
      __syncthreads();
    }
    const bool first = cs == 0, last = cs == NSL - 1;
    for (int ib = slot; ib < pl.n; ib += tpp * BL) {
      PackB<G::VEC> gq[BL];
#pragma unroll
      for (int q = 0; q < BL; ++q)
        gq[q] = ldp<G::VEC>(d_z + (row0 + min(ib + q * tpp, pl.n - 1)) * DF + cs * D + d0);
#pragma unroll
      for (int q = 0; q < BL; ++q) {
        const int i = ib + q * tpp;
        if (i >= pl.n) break;
        const PackB<G::VEC>& g = gq[q];
#pragma unroll
        for (int v = 0; v < G::VEC; ++v) bias_acc[cs].v[v] += g.v[v];
        const float sti = si[i];
        float dot = 0.f;
        for (int r = 0; r < rounds; ++r) {
          const int p = i * pl.pitch + r * 16 + l16;
          const int j = nb_l[p];            // padding = sentinel n (zero row, alpha 0)
          float tsum = 0.f;
          dot_steps<D>(tile_lane, j * (D * 4), g, tsum);
          if constexpr (G::NS == 2) tsum += __shfl_xor(tsum, 16);
          if constexpr (NSL > 1) {
            if (!first) tsum += dpi_t[p];    // the earlier slices' share, written by this lane
          }
          if (last) dot = fmaf(al_t[p], tsum, dot);
          if (slice == 0) dpi_t[p] = tsum;   // d_alpha for now; finished below
        }
        if (!last) continue;
        dot = row16_sum(dot);
        float dsi = 0.f;
        if (slice == 0) {
          for (int r = 0; r < rounds; ++r) {
            const int p = i * pl.pitch + r * 16 + l16;
            const int j = nb_l[p];
            const float de = al_t[p] * (dpi_t[p] - dot);   // dpi_t[p] written by this lane above
            const float pi = sti + sj[j];
            const float dpi = de * (pi > 0.f ? 1.f : GDN_NEG_SLOPE);
            dpi_t[p] = dpi;
            dsi += pi > 0.f ? 0.f : de;       // see below: only the negative-logit slots are summed
          }
        }
        // sum_p slope_p de_p = (0.2 - 1) * sum over the slots with a negative logit (the de_p of a target sum to
        // 0): the cancelling slope-1 terms are never added, a target with positive logits only gets exactly 0
        dsi = row16_sum(dsi) * (GDN_NEG_SLOPE - 1.f);
        if (l16 == 0 && slice == 0) d_si[row0 + i] = dsi;
      }
    }
    }
#pragma unroll
    for (int cs = 0; cs < NSL; ++cs) {
    __syncthreads();
    if constexpr (DX) {   // the tile now holds (this column slice of) d_z of this window (row n stays 0)
      stage_tile(d_z + row0 * DF, cs * D);
      __syncthreads();
    }

    // ---- pass 2: per source, over its reverse list; the first two rounds of BL sources are fetched together
    for (int jb = slot; jb < pl.n; jb += tpp * BL) {
      int lenq[BL];
      uint32_t e0[BL], e1[BL];
#pragma unroll
      for (int q = 0; q < BL; ++q) {
        const int j = min(jb + q * tpp, pl.n - 1);
        const uint32_t* rrow = rent + (size_t)j * pl.rpitch;
        lenq[q] = rlen[j];
        e0[q] = rrow[l16];                                   // rpitch >= 16: always in bounds
        e1[q] = rrow[pl.rpitch >= 32 ? 16 + l16 : l16];
      }
#pragma unroll
      for (int q = 0; q < BL; ++q) {
        const int j = jb + q * tpp;
        if (j >= pl.n) break;
        const int len = lenq[q];
        const uint32_t* rrow = rent + (size_t)j * pl.rpitch;
        PackB<G::VEC> acc;
#pragma unroll
        for (int v = 0; v < G::VEC; ++v) acc.v[v] = 0.f;
        float dsj = 0.f;
        for (int r0 = 0; r0 < len; r0 += 16) {
          const int e = r0 + l16;
          const bool valid = e < len;
          const uint32_t ent = r0 == 0 ? e0[q] : (r0 == 16 ? e1[q] : rrow[valid ? e : 0]);
          const int i = valid ? (int)(ent >> 16) : 0, p = valid ? (int)(ent & 0xffff) : 0;
          if (cs == 0) dsj += valid ? dpi_t[i * pl.pitch + p] : 0.f;
          if constexpr (DX) {
            const float a = valid ? al_t[i * pl.pitch + p] : 0.f;
            axpy_steps<D>(tile_lane, a, (valid ? i : pl.n) * (D * 4), acc);
          }
        }
        if constexpr (DX) stp<G::VEC>(d_xlin + (row0 + j) * DF + cs * D + d0, acc);
        if (cs == 0) {
          dsj = row16_sum(dsj);
          if (l16 == 0 && slice == 0) d_sj[row0 + j] = dsj;
        }
      }
    }
    }
    __syncthreads();   // tables and tile are rewritten by the next window
  }
  // d_bias = column sums of d_z: the lane groups' sums meet in LDS in a fixed order (the tile is free by now),
  // the workgroups' rows in gdn_colsum_ticket — no floating-point atomics, bitwise reproducible
  float* gsum = smem + pl.off_red;                      // [tpp][DF] then nth floats (the tile itself when it is large enough)
#pragma unroll
  for (int s = 0; s < NSL; ++s) stp<G::VEC>(gsum + slot * DF + s * D + d0, bias_acc[s]);
  __syncthreads();
  for (int t = tid; t < DF; t += nth) {
    float s = 0.f;
    for (int q = 0; q < tpp; ++q) s += gsum[q * DF + t];
    dbias[t] = s;
  }
  __syncthreads();
  gdn_colsum_ticket(bias_ws, dbias, DF, d_bias, gsum + tpp * DF);
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

// d_lin_w[d,w] = sum_rows d_xlin[row,d] x[row,w];  d_a[2,64] = sum_rows d_s[row] x[row,:];
// d_c[2,n] = sum_b d_s[b*n + s].
//
// A thread keeps ONE d and ALL 16 columns of a pass in registers across every window of the workgroup, for
// the rows of its row group (256/D groups take the rows of a chunk round robin): per row it reads
// d_xlin[row,d] once (consecutive lanes = consecutive d, conflict free) and the 16 x values of the row as four
// broadcast ds_read_b128 — 5 LDS instructions per 16 FMAs.  (Round 1 owned D/16 columns per thread: 5 LDS
// instructions per 4 FMAs, and the kernel sat at 18 us of a 190 us training step, LDS-issue bound; an earlier
// version still accumulated lane groups into LDS with ds_add_f32 and spent 2/3 of its time there.)  The row
// groups' partial sums meet once, at the end of a pass, through LDS in a fixed order.  A pass covers 16
// columns; w > 16 takes wp/16 passes.  Each workgroup ends with ONE partial row [D*wp + 128 + 2n];
// gdn_project_reduce_kernel sums the rows.
template <int D>
__global__ __launch_bounds__(256) void gdn_project_bwd_kernel(
    int batch, int n, int w, int wp, int rc, const float* __restrict__ x, const float* __restrict__ d_xlin,
    const float* __restrict__ d_si, const float* __restrict__ d_sj, float* __restrict__ part) {
  constexpr int RG = 256 / D;        // row groups
  extern __shared__ float4 smem_b4[];
  float* smem = reinterpret_cast<float*>(smem_b4);
  float* gs = smem;                      // [rc][D]   d_xlin rows of the chunk
  float* xs = gs + (size_t)rc * D;       // [rc][wp]  x rows, zero padded to wp columns
  float* ds = xs + (size_t)rc * wp;      // [2][rc]   d_si, d_sj
  // [2][n] sums over this workgroup's windows, behind the staging area AND behind the 16 x 256 floats the
  // end-of-pass reduction borrows from it (small n: the staging area alone is shorter than that)
  float* dc = smem + max(rc * (wp + D + 2), 16 * 256);
  const int tid = threadIdx.x;
  const int d = tid % D, rg = tid / D;
  const int nout = D * wp;
  float* row_out = part + (size_t)blockIdx.x * (nout + 128 + 2 * n);
  const int a_which = tid >> 6, a_c = tid & 63;          // d_a owner (tid < 128)
  const bool a_owner = tid < 128 && a_c < wp;

  for (int t = tid; t < 2 * n; t += 256) dc[t] = 0.f;

  for (int pass = 0; pass * 16 < wp; ++pass) {
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float acc_a = 0.f;
    const int c0 = pass * 16;
    for (int b = blockIdx.x; b < batch; b += gridDim.x) {
      const size_t row0 = (size_t)b * n;
      for (int r0 = 0; r0 < n; r0 += rc) {
        const int rows = min(rc, n - r0);
        __syncthreads();   // previous chunk fully consumed
        {
          // eight unconditional (clamped) loads per thread are issued before the first LDS store, so a
          // chunk costs one or two HBM round trips instead of one per loop iteration
          const float4* src = reinterpret_cast<const float4*>(d_xlin + (row0 + r0) * D);
          float4* dst = reinterpret_cast<float4*>(gs);
          const int tot4 = rows * D / 4;
          for (int base = 0; base < tot4; base += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(base + u * 256 + tid, tot4 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (base + u * 256 + tid < tot4) dst[base + u * 256 + tid] = v[u];
          }
          const float* xg = x + (row0 + r0) * w;
          const int totx = rows * wp;
          for (int base = 0; base < totx; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int t = min(base + u * 256 + tid, totx - 1);
              const int r = t / wp, c = t - r * wp;
              v[u] = xg[(size_t)r * w + min(c, w - 1)];
              if (c >= w) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (base + u * 256 + tid < totx) xs[base + u * 256 + tid] = v[u];
          }
          for (int t = tid; t < rows; t += 256) {
            const float a = d_si[row0 + r0 + t], bq = d_sj[row0 + r0 + t];
            ds[t] = a;
            ds[rc + t] = bq;
            if (pass == 0) {
              dc[r0 + t] += a;
              dc[n + r0 + t] += bq;
            }
          }
        }
        __syncthreads();
#pragma unroll 4
        for (int r = rg; r < rows; r += RG) {
          const float g = gs[r * D + d];
          const float* xr = xs + r * wp + c0;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + min(4 * q4, wp - c0 - 4));   // wp = 8: columns 8.. repeat 4..7, never stored
            acc[4 * q4 + 0] = fmaf(g, xv.x, acc[4 * q4 + 0]);
            acc[4 * q4 + 1] = fmaf(g, xv.y, acc[4 * q4 + 1]);
            acc[4 * q4 + 2] = fmaf(g, xv.z, acc[4 * q4 + 2]);
            acc[4 * q4 + 3] = fmaf(g, xv.w, acc[4 * q4 + 3]);
          }
        }
        if (pass == 0 && a_owner) {
          const float* dsel = ds + a_which * rc;
#pragma unroll 4
          for (int r = 0; r < rows; ++r) acc_a = fmaf(dsel[r], xs[r * wp + a_c], acc_a);
        }
      }
    }
    // the row groups' partial sums of this pass: through LDS (gs is free: every chunk has been consumed)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) gs[(rg * 16 + q) * D + d] = acc[q];
    __syncthreads();
    for (int o = tid; o < 16 * D; o += 256) {
      const int q = o / D, dd = o - q * D;
      const int c = pass * 16 + q;
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < RG; ++g2) t += gs[(g2 * 16 + q) * D + dd];
      if (c < wp) row_out[c * D + dd] = t;
    }
    if (pass == 0 && tid < 128) row_out[nout + tid] = a_owner ? acc_a : 0.f;
  }
  __syncthreads();
  for (int t = tid; t < 2 * n; t += 256) row_out[nout + 128 + t] = dc[t];
}

// out = sum over the workgroups' partial rows [rows][D*wp + 128 + 2n] -> d_lin_w[D,w], d_a[2,64], d_c[2,n].
// A workgroup sums 16 columns: 16 lane groups take every 16th row (four independent chains each, so the
// loads overlap), then an LDS reduce.
__global__ __launch_bounds__(256) void gdn_project_reduce_kernel(const float* __restrict__ part, int rows, int d,
                                                                 int n, int w, int wp,
                                                                 float* __restrict__ d_lin_w,
                                                                 float* __restrict__ d_a,
                                                                 float* __restrict__ d_c) {
  gdn_project_reduce_body(part, rows, d, n, w, wp, d_lin_w, d_a, d_c, (int)blockIdx.x);
}

// Chain rule through the folded constants a = lin^T att (node_terms) and c = emb . att_em:
// d_lin_w += att_i (x) d_a[0] + att_j (x) d_a[1];  d_att = lin_w d_a;  d_att_em = emb^T d_c;
// d_emb = d_c[0] (x) att_em_i + d_c[1] (x) att_em_j.  Tiny ([d,w], [n,d]); one launch instead of a dozen.
__global__ __launch_bounds__(1024) void gdn_terms_bwd_kernel(
    const float* __restrict__ lin_w, const float* __restrict__ att_i, const float* __restrict__ att_j,
    const float* __restrict__ att_em_i, const float* __restrict__ att_em_j, const float* __restrict__ emb,
    const float* __restrict__ d_a, const float* __restrict__ d_c, int n, int d, int w,
    float* __restrict__ d_lin_w, float* __restrict__ d_att_i, float* __restrict__ d_att_j,
    float* __restrict__ d_att_em_i, float* __restrict__ d_att_em_j, float* __restrict__ d_emb,
    int accumulate_emb) {
  const int tid = threadIdx.x;
  constexpr int NT = 1024;   // block 0's column sums are chains of dependent-latency loads: 16 row groups, not 4
  for (int t = blockIdx.x * NT + tid; t < n * d; t += gridDim.x * NT) {
    const int s = t / d, c = t - s * d;
    const float v = fmaf(d_c[s], att_em_i[c], d_c[n + s] * att_em_j[c]);
    d_emb[t] = accumulate_emb ? d_emb[t] + v : v;   // (+ the head's share, written there by gdn_head_train_bwd)
  }
  if (blockIdx.x != 0) return;
  __shared__ float part[4][NT];
  for (int t = tid; t < d * w; t += NT) {
    const int c = t / w, q = t - c * w;
    d_lin_w[t] += fmaf(att_i[c], d_a[q], att_j[c] * d_a[GDN_A_PITCH + q]);
  }
  // column c is handled by the 256/d threads tid = c, c+d, ...: strided partial sums, then an LDS reduce
  const int c = tid % d, g = tid / d, groups = NT / d;
  float si = 0.f, sj = 0.f, ei = 0.f, ej = 0.f;
  // (independent loads, eight iterations in flight: left rolled, the two loops were ~35 dependent L2 round
  // trips and the kernel took 15 us of a 218 us training step)
#pragma unroll 8
  for (int q = g; q < w; q += groups) {
    const float lw = lin_w[c * w + q];
    si = fmaf(lw, d_a[q], si);
    sj = fmaf(lw, d_a[GDN_A_PITCH + q], sj);
  }
#pragma unroll 8
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
  return min(batch, gdn_cu_count() * gdn_blocks_per_cu(reinterpret_cast<const void*>(kern), threads, lds));
}

#define GDN_PBWD_MAX_ROWS 1024   // partial rows (= workgroups) of gdn_project_bwd

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

// LDS plan of the backward: with_tables = the two [n, pitch] tables and the lists sit beside the tile;
// td = columns of the tile (d, or 64 when the kernel walks column slices)
static bool bwd_plan(int batch, int n, int d, int k, bool with_tables, BwdPlan* pl, int td = 0) {
  if (td <= 0) td = d;
  pl->n = n; pl->d = d; pl->k = k; pl->batch = batch; pl->pitch = gdn_nbr_pitch(k); pl->rpitch = gdn_rev_pitch(n);
  const int npad = (n + 1 + 3) & ~3;   // +1: the sentinel index n used as list padding
  int off = 0;
  pl->off_tile = off; off += (n + 1) * td;
  pl->off_sj = off; off += npad;
  pl->off_si = off; off += npad;
  pl->off_al = off; off += with_tables ? n * pl->pitch : 0;
  pl->off_dpi = off; off += with_tables ? n * pl->pitch : 0;
  pl->off_dbias = off; off += d;
  pl->off_nbr = off; off += with_tables ? n * pl->pitch / 2 : 0;   // u16 lists
  // end-of-kernel d_bias reduction: [lane groups <= 32][d] partial rows + one float per thread (<= 512); it
  // borrows the tile when that is large enough (every window is done by then)
  const int red = 32 * (td < d ? d : (d < 64 ? d : 64)) + 512;
  if ((n + 1) * td >= red) pl->off_red = pl->off_tile;
  else { pl->off_red = off; off += red; }
  pl->lds_bytes = off * 4;
  return pl->lds_bytes <= 160 * 1024;
}

// GDN_BWD_SLICED=1 (read once per process): d = 128 always walks two 64-column slices (A/B against the whole tile)
static bool gdn_bwd_sliced_forced() { return GDN_ENV_INT_ONCE("GDN_BWD_SLICED", 0) != 0; }

// workspace = [ticket + GDN_COLSUM_MAX_ROWS partial rows of d_bias][d_pi tables when they do not fit LDS]
static long long bwd_bias_ws_floats(int d) { return GDN_COLSUM_WS_HEAD + (long long)GDN_COLSUM_MAX_ROWS * d; }

extern "C" long long gdn_attn_aggregate_bwd_workspace_bytes(int batch, int n, int d, int k) {
  if (batch <= 0 || n <= 0 || k <= 0 || k > n || d <= 0) return 0;
  BwdPlan pl;
  long long floats = bwd_bias_ws_floats(d);
  if (!bwd_plan(batch, n, d, k, true, &pl) || (d == 128 && gdn_bwd_sliced_forced()))
    floats += (long long)batch * n * gdn_nbr_pitch(k);   // tables beyond LDS
  return floats * (long long)sizeof(float);
}

// GDN_BWD_PATH=valu (read once per process) keeps the row-gather backward at every shape (A/B runs)
static bool gdn_bwd_valu_forced() {
  static const bool valu_bwd = [] { const char* e = getenv("GDN_BWD_PATH"); return e && e[0] == 'v'; }();
  return valu_bwd;
}

// 1 when gdn_attn_aggregate_bwd reads the reverse lists (gdn_graph_reverse) at this shape, 0 when it runs
// the matrix-core backward, which does not (rent / rlen may then be null and the launch can be skipped)
extern "C" int gdn_attn_aggregate_bwd_uses_reverse(int n, int d, int k) {
  if (n <= 0 || k <= 0 || k > n) return 1;
  return (d == 64 && !gdn_bwd_valu_forced() && gdn_use_dense_path() && gdn_dense_supported(n, 1, d, k)) ? 0 : 1;
}

static int attn_aggregate_bwd_impl(const float* d_z, const float* xlin, const float* alpha,
                                   const float* s_i, const float* s_j, const uint16_t* nbr,
                                   const uint32_t* rent, const int32_t* rlen, int batch, int n, int d,
                                   int k, float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                                   float* workspace, void* stream, bool wide) {
  const bool dense = !wide && !gdn_attn_aggregate_bwd_uses_reverse(n, d, k);
  if (!d_z || !xlin || !alpha || !s_i || !s_j || !nbr || !d_xlin || !d_si || !d_sj || !d_bias || !workspace ||
      batch <= 0 || n <= 0 || k <= 0)
    return GDN_ERR_ARG;
  if ((!rent || !rlen) && !dense) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (k > n || n > 4096 || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
  BwdPlan pl;
  bool glb = false, sliced = false;
  if (gdn_bwd_sliced_forced() && d == 128) {
    if (!bwd_plan(batch, n, d, k, false, &pl, 64)) return GDN_ERR_UNSUPPORTED;
    glb = sliced = true;
  } else if (!bwd_plan(batch, n, d, k, true, &pl)) {
    // tables through global memory (the workspace behind the d_bias rows); the tile alone must still fit,
    // whole or (d = 128) as two 64-column slices the workgroup walks one after the other
    glb = true;
    if (!bwd_plan(batch, n, d, k, false, &pl)) {
      if (d != 128 || !bwd_plan(batch, n, d, k, false, &pl, 64)) return GDN_ERR_UNSUPPORTED;
      sliced = true;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  float* dpi_ws = workspace + bwd_bias_ws_floats(d);
  const bool many_rows = n * (d / 64 > 0 ? d / 64 : 1) > 64;   // enough rows for 32 lane groups
  // matrix-core shapes: both halves of the backward as dense products (gdn_forward_dense.hip)
  if (dense)
    return gdn_dense_attn_bwd(d_z, xlin, alpha, s_i, s_j, nbr, batch, n, k, d_xlin, d_si, d_sj, d_bias, workspace, st);
#define GDN_BWD_NT(DD, NT, GL)                                                                        \
  {                                                                                                   \
    const int grid = min(occupancy_grid(gdn_attn_bwd_kernel<DD, NT, GL>, NT, pl.lds_bytes, batch),    \
                         GDN_COLSUM_MAX_ROWS);                                                        \
    hipLaunchKernelGGL((gdn_attn_bwd_kernel<DD, NT, GL>), dim3(grid), dim3(NT), pl.lds_bytes, st, pl, d_z, xlin, \
                       alpha, s_i, s_j, nbr, rent, rlen, d_xlin, d_si, d_sj, d_bias, dpi_ws, workspace); \
  }
#define GDN_BWD(DD)                                                   \
  case DD:                                                            \
    if (glb) GDN_BWD_NT(DD, 512, true)                                \
    else if (many_rows) GDN_BWD_NT(DD, 512, false)                         \
    else GDN_BWD_NT(DD, 256, false)                                   \
    break;
  if (sliced) {
    const int grid = min(occupancy_grid(gdn_attn_bwd_kernel<64, 512, true, true, 2>, 512, pl.lds_bytes, batch),
                         GDN_COLSUM_MAX_ROWS);
    hipLaunchKernelGGL((gdn_attn_bwd_kernel<64, 512, true, true, 2>), dim3(grid), dim3(512), pl.lds_bytes, st, pl,
                       d_z, xlin, alpha, s_i, s_j, nbr, rent, rlen, d_xlin, d_si, d_sj, d_bias, dpi_ws, workspace);
    return gdn_launch_status();
  }
  switch (d) {
    GDN_BWD(16)
    GDN_BWD(32)
    GDN_BWD(64)
    GDN_BWD(128)
  }
#undef GDN_BWD
#undef GDN_BWD_NT
  return gdn_launch_status();
}

extern "C" int gdn_attn_aggregate_bwd(const float* d_z, const float* xlin, const float* alpha,
                                      const float* s_i, const float* s_j, const uint16_t* nbr,
                                      const uint32_t* rent, const int32_t* rlen, int batch, int n, int d,
                                      int k, float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                                      float* workspace, void* stream) {
  return attn_aggregate_bwd_impl(d_z, xlin, alpha, s_i, s_j, nbr, rent, rlen, batch, n, d, k, d_xlin, d_si, d_sj,
                                 d_bias, workspace, stream, false);
}
// `_wide`: the row-gather backward at every shape (reverse lists required)
extern "C" int gdn_attn_aggregate_bwd_wide(const float* d_z, const float* xlin, const float* alpha,
                                           const float* s_i, const float* s_j, const uint16_t* nbr,
                                           const uint32_t* rent, const int32_t* rlen, int batch, int n, int d,
                                           int k, float* d_xlin, float* d_si, float* d_sj, float* d_bias,
                                           float* workspace, void* stream) {
  return attn_aggregate_bwd_impl(d_z, xlin, alpha, s_i, s_j, nbr, rent, rlen, batch, n, d, k, d_xlin, d_si, d_sj,
                                 d_bias, workspace, stream, true);
}

// 1 when every kernel of a training step (staged forward, this file's backward) takes the shape: what
// harness.NativeTrainStep.applicable() asks before it commits to the captured step
extern "C" int gdn_train_supported(int n, int w, int d, int k) {
  if (n <= 0 || w <= 0 || k <= 0 || k > n || n > 4096 || k + 1 > 1024 || w > GDN_MAX_W) return 0;
  if (d != 16 && d != 32 && d != 64 && d != 128) return 0;
  BwdPlan pl;
  if (!bwd_plan(1, n, d, k, true, &pl) && !bwd_plan(1, n, d, k, false, &pl) &&
      !(d == 128 && bwd_plan(1, n, d, k, false, &pl, 64)))
    return 0;
  const int wp = w <= 8 ? 8 : ((w + 15) & ~15);
  int rc = (24576 - 2 * n) / (wp + d + 2);                  // gdn_project_bwd's staging chunk
  if (rc < 1) return 0;
  return gdn_forward_staged_ok(n, w, d, k);
}

extern "C" long long gdn_project_bwd_workspace_bytes(int n, int w, int d) {
  if (n <= 0 || w <= 0 || d <= 0) return 0;
  const int wp = w <= 8 ? 8 : ((w + 15) & ~15);
  return (long long)GDN_PBWD_MAX_ROWS * (d * wp + 128 + 2 * n) * (long long)sizeof(float);
}

// first half of gdn_project_bwd: the per-workgroup partial rows; *rows_out = how many there are
extern "C" int gdn_project_bwd_partials(const float* x, const float* d_xlin, const float* d_si, const float* d_sj,
                                        int batch, int n, int w, int d, float* workspace, int* rows_out,
                                        void* stream) {
  if (!x || !d_xlin || !d_si || !d_sj || !workspace || !rows_out || batch <= 0 || n <= 0 || w <= 0) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (w > GDN_MAX_W || n > 4096) return GDN_ERR_UNSUPPORTED;
  const int wp = w <= 8 ? 8 : ((w + 15) & ~15);
  // rows per staged chunk: whole window when it fits ~96 KB (three workgroups per CU at the SWaT shape)
  int rc = (24576 - 2 * n) / (wp + d + 2);
  if (rc > n) rc = n;
  if (rc < 1) return GDN_ERR_UNSUPPORTED;
  const int lds = (max(rc * (wp + d + 2), 16 * 256) + 2 * n) * 4;
  hipStream_t st = (hipStream_t)stream;
  int grid = 1;
#define GDN_PB(DD)                                                                                  \
  case DD: {                                                                                        \
    grid = min(occupancy_grid(gdn_project_bwd_kernel<DD>, 256, lds, batch), GDN_PBWD_MAX_ROWS);     \
    hipLaunchKernelGGL((gdn_project_bwd_kernel<DD>), dim3(grid), dim3(256), lds, st, batch, n, w, wp, rc, \
                       x, d_xlin, d_si, d_sj, workspace);                                           \
  } break;
  switch (d) {
    GDN_PB(16)
    GDN_PB(32)
    GDN_PB(64)
    GDN_PB(128)
  }
#undef GDN_PB
  *rows_out = grid;
  return gdn_launch_status();
}

extern "C" int gdn_project_bwd(const float* x, const float* d_xlin, const float* d_si, const float* d_sj,
                               int batch, int n, int w, int d, float* workspace, float* d_lin_w, float* d_a,
                               float* d_c, void* stream) {
  if (!d_lin_w || !d_a || !d_c) return GDN_ERR_ARG;
  int rows = 0;
  const int rc = gdn_project_bwd_partials(x, d_xlin, d_si, d_sj, batch, n, w, d, workspace, &rows, stream);
  if (rc != GDN_OK) return rc;
  const int wp = w <= 8 ? 8 : ((w + 15) & ~15);
  const int len = d * wp + 128 + 2 * n;
  hipLaunchKernelGGL(gdn_project_reduce_kernel, dim3((len + 15) / 16), dim3(256), 0, (hipStream_t)stream, workspace,
                     rows, d, n, w, wp, d_lin_w, d_a, d_c);
  return gdn_launch_status();
}

extern "C" int gdn_terms_bwd_acc(const float* lin_w, const float* att_i, const float* att_j,
                             const float* att_em_i, const float* att_em_j, const float* emb,
                             const float* d_a, const float* d_c, int n, int d, int w, float* d_lin_w,
                             float* d_att_i, float* d_att_j, float* d_att_em_i, float* d_att_em_j,
                             float* d_emb, int accumulate_emb, void* stream) {
  if (!lin_w || !att_i || !att_j || !att_em_i || !att_em_j || !emb || !d_a || !d_c || !d_lin_w || !d_att_i ||
      !d_att_j || !d_att_em_i || !d_att_em_j || !d_emb || n <= 0 || d <= 0 || w <= 0)
    return GDN_ERR_ARG;
  if (w > GDN_MAX_W || d > 256 || (256 % d) != 0) return GDN_ERR_UNSUPPORTED;
  int grid = (n * d + 1024 * 2 - 1) / (1024 * 2);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(gdn_terms_bwd_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)stream, lin_w, att_i, att_j,
                     att_em_i, att_em_j, emb, d_a, d_c, n, d, w, d_lin_w, d_att_i, d_att_j, d_att_em_i,
                     d_att_em_j, d_emb, accumulate_emb);
  return gdn_launch_status();
}

extern "C" int gdn_terms_bwd(const float* lin_w, const float* att_i, const float* att_j,
                             const float* att_em_i, const float* att_em_j, const float* emb,
                             const float* d_a, const float* d_c, int n, int d, int w, float* d_lin_w,
                             float* d_att_i, float* d_att_j, float* d_att_em_i, float* d_att_em_j,
                             float* d_emb, void* stream) {
  return gdn_terms_bwd_acc(lin_w, att_i, att_j, att_em_i, att_em_j, emb, d_a, d_c, n, d, w, d_lin_w, d_att_i, d_att_j,
                           d_att_em_i, d_att_em_j, d_emb, 0, stream);
}
