// GDN forward on gfx950: projection, attention + softmax + neighbour aggregation, output head.
//
// Work decomposition (all three kernels): ONE WORKGROUP = ONE WINDOW (n sensors), looping over
// windows b = blockIdx.x, blockIdx.x + gridDim.x, ...  The window's projected features
// xlin[n, d] live in an LDS tile; the neighbour lists (shared by every window) are staged
// into LDS once per workgroup.
//
// Lane mapping: a wave is four 16-lane DPP rows.  A row owns one target sensor at a time and
// its 16 lanes own 16 consecutive VEC-wide column chunks of a 64-column slice (d = 128: two
// rows per target, one per slice).  Source rows are fetched from the LDS tile with one
// ds_read_b128 per lane per neighbour: 4 targets x 256 B per wave-instruction, bank-conflict
// free for any set of rows because a lane's bank group depends only on its column chunk.
//
// The per-target softmax puts neighbour p on lane p%16 of the row; max and sum are DPP
// butterflies.  During accumulation the (alpha, row-offset) pair held by each lane is
// ROTATED through the row with DPP row_ror:1, so every lane meets every neighbour after 16
// steps without any LDS broadcast traffic: lane l adds neighbour (p+s)%16 at step s — the
// order of the sum differs per lane, the set does not.
#include "gdn_common.hpp"

namespace {

template <int V>
struct Pack {
  float v[V];
};

template <int V>
__device__ __forceinline__ Pack<V> ld_pack(const float* p) {
  Pack<V> r;
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
__device__ __forceinline__ void st_pack(float* p, const Pack<V>& r) {
  if constexpr (V == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else if constexpr (V == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  } else {
    *p = r.v[0];
  }
}

template <int D>
struct Geo {
  static constexpr int DS = D < 64 ? D : 64;  // columns per 16-lane row
  static constexpr int VEC = DS / 16;         // columns per lane
  static constexpr int NS = D / DS;           // rows (slices) per target
};

// LDS carve-up, in floats from the 16-B aligned dynamic base.  Host and device agree via Plan.
struct Plan {
  int n, d, w, wp, k, pitch, batch;
  int xrows;     // rows of x staged per chunk (project / fused)
  int nbr_lds;   // 1: neighbour lists in LDS, 0: read through L2
  int off_xl, off_si, off_sj, off_deg, off_nbr, off_xs;  // float offsets
  int lds_bytes;
};

enum { MODE_PROJECT = 0, MODE_ATTN = 1, MODE_FUSED = 2 };

struct Args {
  // inputs
  const float* x;           // [B, n, w]
  const float* lin_w;       // [d, w]
  const float* node_terms;  // [a_i(64) | a_j(64) | c_i(n) | c_j(n)]
  const float* xlin_in;     // [BN, d]  (MODE_ATTN)
  const float* si_in;       // [BN]
  const float* sj_in;       // [BN]
  const uint16_t* nbr;      // [n, pitch]
  const int32_t* deg;       // [n]
  const float* gnn_bias;    // [d]
  const float* emb;         // [n, d]
  const float* bn1;         // [scale(d) | shift(d)]
  const float* bn2;
  const float* out_w;       // [d]
  const float* out_b;       // [1]
  // outputs
  float* xlin_out;  // [BN, d]   (MODE_PROJECT)
  float* si_out;
  float* sj_out;
  float* z;         // [BN, d]   (MODE_ATTN)
  float* alpha;     // [BN, pitch] or null
  float* out;       // [BN]      (MODE_FUSED)
};

// ------------------------------------------------------------------ projection phase
// This lane's [VEC x WCH] block of lin.weight for W-chunk wc (zero beyond w).
template <int D, int WCH>
__device__ __forceinline__ void load_lane_weights(const Plan& pl, const Args& a, int wc,
                                                  float (&wl)[Geo<D>::VEC][WCH]) {
  using G = Geo<D>;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int d0 = (grp % G::NS) * 64 + l16 * G::VEC;
#pragma unroll
  for (int v = 0; v < G::VEC; ++v)
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
      const int col = wc * WCH + c;
      wl[v][c] = col < pl.w ? a.lin_w[(size_t)(d0 + v) * pl.w + col] : 0.f;
    }
}

// rows [r0, r1) of window b, W-chunk wc of nch: xs (LDS, pitch wp) -> xlin tile (LDS, and global
// when TO_GLOBAL) and the attention scalars s_i / s_j.
template <int D, int WCH, bool TO_GLOBAL>
__device__ __forceinline__ void project_chunk(const Plan& pl, const Args& a, float* smem, int b, int r0,
                                              int r1, int wc, int nch,
                                              const float (&wl)[Geo<D>::VEC][WCH]) {
  using G = Geo<D>;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (blockDim.x >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  float* xl = smem + pl.off_xl;
  float* si = smem + pl.off_si;
  float* sj = smem + pl.off_sj;
  const float* xs = smem + pl.off_xs;
  // this lane's share of a_i / a_j for the 16-lane dot product
  const int acol = wc * WCH + l16;
  const bool ahas = l16 < WCH;
  const float ai = ahas ? a.node_terms[acol] : 0.f;
  const float aj = ahas ? a.node_terms[GDN_A_PITCH + acol] : 0.f;
  const bool last = wc == nch - 1;

  for (int row = r0 + slot; row < r1; row += tpp) {
    const float* xrow = xs + (size_t)(row - r0) * pl.wp + wc * WCH;
    float xr[WCH];
#pragma unroll
    for (int c = 0; c < WCH; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(xrow + c);
      xr[c] = t.x; xr[c + 1] = t.y; xr[c + 2] = t.z; xr[c + 3] = t.w;
    }
    // attention scalars: x_row . a  (16-lane dot, DPP butterfly)
    const float xv = ahas ? xrow[l16] : 0.f;
    float pi = row16_sum(xv * ai);
    float pj = row16_sum(xv * aj);
    if (slice == 0 && l16 == 0) {
      if (wc == 0) {
        pi += a.node_terms[2 * GDN_A_PITCH + row];
        pj += a.node_terms[2 * GDN_A_PITCH + pl.n + row];
      } else {
        pi += si[row];
        pj += sj[row];
      }
      si[row] = pi;
      sj[row] = pj;
      if (TO_GLOBAL && last) {
        a.si_out[(size_t)b * pl.n + row] = pi;
        a.sj_out[(size_t)b * pl.n + row] = pj;
      }
    }
    Pack<G::VEC> acc;
    if (wc == 0) {
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) acc.v[v] = 0.f;
    } else {
      acc = ld_pack<G::VEC>(xl + (size_t)row * D + d0);
    }
#pragma unroll
    for (int c = 0; c < WCH; ++c)
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) acc.v[v] = fmaf(xr[c], wl[v][c], acc.v[v]);
    st_pack<G::VEC>(xl + (size_t)row * D + d0, acc);
    if (TO_GLOBAL && last) st_pack<G::VEC>(a.xlin_out + ((size_t)b * pl.n + row) * D + d0, acc);
  }
}

// ------------------------------------------------------------------ attention + aggregation
template <int D, int MODE>
__device__ __forceinline__ void aggregate_window(const Plan& pl, const Args& a, float* smem, int b) {
  using G = Geo<D>;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (blockDim.x >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  const float* xl = smem + pl.off_xl;
  const float* si = smem + pl.off_si;
  const float* sj = smem + pl.off_sj;
  const uint16_t* degs = reinterpret_cast<const uint16_t*>(smem + pl.off_deg);
  const uint16_t* nbr = pl.nbr_lds ? reinterpret_cast<const uint16_t*>(smem + pl.off_nbr) : a.nbr;
  const char* xl_lane = reinterpret_cast<const char*>(xl + d0);

  // per-lane constants of the epilogue
  Pack<G::VEC> bias = ld_pack<G::VEC>(a.gnn_bias + d0);
  Pack<G::VEC> sc1, sh1, sc2, sh2, wo;
  float out_b = 0.f;
  if constexpr (MODE == MODE_FUSED) {
    sc1 = ld_pack<G::VEC>(a.bn1 + d0);
    sh1 = ld_pack<G::VEC>(a.bn1 + D + d0);
    sc2 = ld_pack<G::VEC>(a.bn2 + d0);
    sh2 = ld_pack<G::VEC>(a.bn2 + D + d0);
    wo = ld_pack<G::VEC>(a.out_w + d0);
    out_b = a.out_b[0];
  }

  for (int i = slot; i < pl.n; i += tpp) {
    const int degi = degs[i];
    const int rounds = (degi + 15) >> 4;
    const float sti = si[i];
    const uint16_t* nrow = nbr + (size_t)i * pl.pitch;
    Pack<G::VEC> emb_i;
    if constexpr (MODE == MODE_FUSED) emb_i = ld_pack<G::VEC>(a.emb + (size_t)i * D + d0);

    // pass 1: row max of LeakyReLU(s_i[i] + s_j[src])
    float m = -INFINITY;
    for (int r = 0; r < rounds; ++r) {
      const int p = r * 16 + l16;
      const float e = leaky(sti + sj[nrow[p]]);
      m = fmaxf(m, p < degi ? e : -INFINITY);
    }
    m = row16_max(m);
    // pass 2: denominator
    float sum = 0.f;
    for (int r = 0; r < rounds; ++r) {
      const int p = r * 16 + l16;
      const float e = leaky(sti + sj[nrow[p]]);
      sum += p < degi ? expf(e - m) : 0.f;
    }
    sum = row16_sum(sum);
    const float inv = 1.f / (sum + GDN_SOFTMAX_EPS);

    // pass 3: alpha-weighted sum of source rows
    Pack<G::VEC> acc;
#pragma unroll
    for (int v = 0; v < G::VEC; ++v) acc.v[v] = 0.f;
    for (int r = 0; r < rounds; ++r) {
      const int p = r * 16 + l16;
      const int j = nrow[p];
      const float e = leaky(sti + sj[j]);
      float al = p < degi ? expf(e - m) * inv : 0.f;
      int jb = j * (D * 4);
      if constexpr (MODE == MODE_ATTN) {
        if (a.alpha && slice == 0) a.alpha[((size_t)b * pl.n + i) * pl.pitch + p] = al;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const Pack<G::VEC> src = ld_pack<G::VEC>(reinterpret_cast<const float*>(xl_lane + jb));
#pragma unroll
        for (int v = 0; v < G::VEC; ++v) acc.v[v] = fmaf(al, src.v[v], acc.v[v]);
        al = dpp_f<GDN_DPP_ROR1>(al);
        jb = dpp_i<GDN_DPP_ROR1>(jb);
      }
    }
    if constexpr (MODE == MODE_ATTN) {
      // zero the alpha slots of rounds this target does not use
      if (a.alpha && slice == 0)
        for (int p = rounds * 16 + l16; p < pl.pitch; p += 16)
          a.alpha[((size_t)b * pl.n + i) * pl.pitch + p] = 0.f;
    }

#pragma unroll
    for (int v = 0; v < G::VEC; ++v) acc.v[v] += bias.v[v];
    if constexpr (MODE == MODE_ATTN) {
      st_pack<G::VEC>(a.z + ((size_t)b * pl.n + i) * D + d0, acc);
    } else {
      float part = 0.f;
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) {
        float h = fmaxf(fmaf(acc.v[v], sc1.v[v], sh1.v[v]), 0.f);  // GDN.py:77-79 (eval BN, ReLU)
        h *= emb_i.v[v];                                            // GDN.py:175-176
        h = fmaxf(fmaf(h, sc2.v[v], sh2.v[v]), 0.f);                // GDN.py:178-180
        part = fmaf(h, wo.v[v], part);                              // OutLayer Linear(d->1)
      }
      part = row16_sum(part);
      if constexpr (G::NS == 2) part += __shfl_xor(part, 16);
      if (l16 == 0 && slice == 0) a.out[(size_t)b * pl.n + i] = part + out_b;
    }
  }
}

// ------------------------------------------------------------------ the kernel
// Stage rows [r0, r1) of x[b] into LDS at pitch wp (zero padded columns).
__device__ __forceinline__ void stage_x(const Plan& pl, const float* xg, float* xs, int r0, int r1) {
  const int cnt = (r1 - r0) * pl.wp;
  for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
    const int r = t / pl.wp, c = t - r * pl.wp;
    xs[t] = c < pl.w ? xg[(size_t)(r0 + r) * pl.w + c] : 0.f;
  }
}

template <int D, int WCH, int MODE, int NT>
__global__ __launch_bounds__(NT) void gdn_window_kernel(const Plan pl, const Args a) {
  using G = Geo<D>;
  extern __shared__ float4 smem_f4[];
  float* smem = reinterpret_cast<float*>(smem_f4);
  const int tid = threadIdx.x, nth = blockDim.x;

  // once per workgroup: neighbour lists + degrees into LDS
  if constexpr (MODE != MODE_PROJECT) {
    uint16_t* degs = reinterpret_cast<uint16_t*>(smem + pl.off_deg);
    for (int t = tid; t < pl.n; t += nth) degs[t] = (uint16_t)a.deg[t];
    if (pl.nbr_lds) {
      // n*pitch u16 = n*pitch/8 uint4 (pitch is a multiple of 16)
      const uint4* src = reinterpret_cast<const uint4*>(a.nbr);
      uint4* dst = reinterpret_cast<uint4*>(smem + pl.off_nbr);
      const int nvec = pl.n * pl.pitch / 8;
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
    }
  }

  if constexpr (MODE == MODE_ATTN) {
    for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
      // stage xlin[b] (n*d floats, 16-B vectors) and the node scalars
      const float4* src = reinterpret_cast<const float4*>(a.xlin_in + (size_t)b * pl.n * D);
      float4* dst = reinterpret_cast<float4*>(smem + pl.off_xl);
      const int nvec = pl.n * D / 4;
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
      float* si = smem + pl.off_si;
      float* sj = smem + pl.off_sj;
      for (int t = tid; t < pl.n; t += nth) {
        si[t] = a.si_in[(size_t)b * pl.n + t];
        sj[t] = a.sj_in[(size_t)b * pl.n + t];
      }
      __syncthreads();
      aggregate_window<D, MODE>(pl, a, smem, b);
      __syncthreads();  // the tile is overwritten by the next window
    }
  } else {
    float* xs = smem + pl.off_xs;
    const int nch = pl.wp / WCH;
    if (nch == 1 && pl.xrows >= pl.n) {
      // common case (w <= 16, whole window in one x chunk): lin.weight block stays in registers
      float wl[G::VEC][WCH];
      load_lane_weights<D, WCH>(pl, a, 0, wl);
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        stage_x(pl, a.x + (size_t)b * pl.n * pl.w, xs, 0, pl.n);
        __syncthreads();
        project_chunk<D, WCH, MODE == MODE_PROJECT>(pl, a, smem, b, 0, pl.n, 0, 1, wl);
        __syncthreads();
        if constexpr (MODE == MODE_FUSED) {
          aggregate_window<D, MODE>(pl, a, smem, b);
          __syncthreads();
        }
      }
    } else {
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        for (int r0 = 0; r0 < pl.n; r0 += pl.xrows) {
          const int r1 = min(pl.n, r0 + pl.xrows);
          stage_x(pl, a.x + (size_t)b * pl.n * pl.w, xs, r0, r1);
          __syncthreads();
          for (int wc = 0; wc < nch; ++wc) {
            float wl[G::VEC][WCH];
            load_lane_weights<D, WCH>(pl, a, wc, wl);
            project_chunk<D, WCH, MODE == MODE_PROJECT>(pl, a, smem, b, r0, r1, wc, nch, wl);
          }
          __syncthreads();
        }
        if constexpr (MODE == MODE_FUSED) {
          aggregate_window<D, MODE>(pl, a, smem, b);
          __syncthreads();
        }
      }
    }
  }
}

// ------------------------------------------------------------------ head (staged eval path)
// z[BN, d] -> out[BN]; 16 lanes per row (VEC columns each; d = 128: two passes per lane).
template <int D>
__global__ __launch_bounds__(256) void gdn_head_kernel(const float* __restrict__ z,
                                                       const float* __restrict__ emb,
                                                       const float* __restrict__ bn1,
                                                       const float* __restrict__ bn2,
                                                       const float* __restrict__ out_w,
                                                       const float* __restrict__ out_b, int rows, int n,
                                                       float* __restrict__ out, float* __restrict__ h2) {
  constexpr int CPL = D / 16;  // columns per lane
  const int l16 = threadIdx.x & 15;
  const int d0 = l16 * CPL;
  float sc1[CPL], sh1[CPL], sc2[CPL], sh2[CPL], wo[CPL];
#pragma unroll
  for (int v = 0; v < CPL; ++v) {
    sc1[v] = bn1[d0 + v]; sh1[v] = bn1[D + d0 + v];
    sc2[v] = bn2[d0 + v]; sh2[v] = bn2[D + d0 + v];
    wo[v] = out_w[d0 + v];
  }
  const float ob = out_b[0];
  const int rpb = blockDim.x >> 4;
  for (int row = blockIdx.x * rpb + (threadIdx.x >> 4); row < rows; row += gridDim.x * rpb) {
    const int s = row % n;
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < CPL; ++v) {
      float h = fmaxf(fmaf(z[(size_t)row * D + d0 + v], sc1[v], sh1[v]), 0.f);
      h *= emb[(size_t)s * D + d0 + v];
      h = fmaxf(fmaf(h, sc2[v], sh2[v]), 0.f);
      if (h2) h2[(size_t)row * D + d0 + v] = h;
      part = fmaf(h, wo[v], part);
    }
    part = row16_sum(part);
    if (l16 == 0) out[row] = part + ob;
  }
}

// ------------------------------------------------------------------ host side
int make_plan(int mode, int batch, int n, int w, int d, int k, Plan* pl, int* threads) {
  if (batch <= 0 || n <= 0 || d <= 0) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (n > 4096) return GDN_ERR_UNSUPPORTED;
  pl->n = n; pl->d = d; pl->w = w; pl->k = k; pl->batch = batch;
  pl->wp = 0; pl->pitch = 0; pl->xrows = 0; pl->nbr_lds = 0;
  if (mode != MODE_ATTN) {
    if (w <= 0) return GDN_ERR_ARG;
    if (w > GDN_MAX_W) return GDN_ERR_UNSUPPORTED;
    pl->wp = w <= 8 ? 8 : ((w + 15) & ~15);
  }
  if (mode != MODE_PROJECT) {
    if (k <= 0) return GDN_ERR_ARG;
    if (k > n || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
    pl->pitch = gdn_nbr_pitch(k);
  }
  const int LDS_MAX = 160 * 1024;
  const int npad = (n + 3) & ~3;
  int off = 0;
  pl->off_xl = off; off += n * d;
  pl->off_si = off; off += npad;
  pl->off_sj = off; off += npad;
  pl->off_deg = off; off += (npad / 2 + 3) & ~3;
  pl->off_nbr = off;
  pl->off_xs = off;
  int fixed = off * 4;
  const int nbr_bytes = n * pl->pitch * 2;
  const int xs_full = n * pl->wp * 4;
  if (fixed > LDS_MAX) return GDN_ERR_UNSUPPORTED;
  // neighbour lists go to LDS when they fit beside a useful x chunk
  int remaining = LDS_MAX - fixed;
  if (mode != MODE_PROJECT && nbr_bytes <= remaining - (mode == MODE_FUSED ? 16 * pl->wp * 4 : 0)) {
    pl->nbr_lds = 1;
    pl->off_xs = pl->off_nbr + nbr_bytes / 4;
    remaining -= nbr_bytes;
  }
  if (mode != MODE_ATTN) {
    if (xs_full <= remaining) {
      pl->xrows = n;
    } else {
      pl->xrows = (remaining / (pl->wp * 4)) & ~15;
      if (pl->xrows < 16) return GDN_ERR_UNSUPPORTED;
    }
    remaining -= pl->xrows * pl->wp * 4;
  }
  pl->lds_bytes = LDS_MAX - remaining;
  // small windows: 256 threads and several workgroups per CU; big tiles own the CU -> 1024 threads
  *threads = pl->lds_bytes > 80 * 1024 ? 512 : 256;
  return GDN_OK;
}

template <int D, int WCH, int MODE, int NT>
int launch_window(const Plan& pl, const Args& a, hipStream_t stream) {
  constexpr int threads = NT;
  auto kern = gdn_window_kernel<D, WCH, MODE, NT>;
  static bool attr_set = false;
  static int occ_cache_lds = -1, occ = 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  if (occ_cache_lds != pl.lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, pl.lds_bytes) != hipSuccess ||
        nb <= 0) {
      (void)hipGetLastError();
      nb = 1;
    }
    occ = nb; occ_cache_lds = pl.lds_bytes;
  }
  const int grid = min(pl.batch, gdn_cu_count() * occ);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), pl.lds_bytes, stream, pl, a);
  return gdn_launch_status();
}

template <int MODE>
int dispatch_window(const Plan& pl, const Args& a, int threads, hipStream_t stream) {
#define GDN_CASE(DD)                                                                  \
  case DD:                                                                            \
    if (threads == 256) {                                                             \
      if (MODE == MODE_ATTN || pl.wp != 8) return launch_window<DD, 16, MODE, 256>(pl, a, stream); \
      return launch_window<DD, 8, MODE, 256>(pl, a, stream);                          \
    }                                                                                 \
    if (MODE == MODE_ATTN || pl.wp != 8) return launch_window<DD, 16, MODE, 512>(pl, a, stream);  \
    return launch_window<DD, 8, MODE, 512>(pl, a, stream);
  switch (pl.d) {
    GDN_CASE(16)
    GDN_CASE(32)
    GDN_CASE(64)
    GDN_CASE(128)
  }
#undef GDN_CASE
  return GDN_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int gdn_project_fwd(const float* x, const float* lin_w, const float* node_terms, int batch,
                               int n, int w, int d, float* xlin, float* s_i, float* s_j, void* stream) {
  if (!x || !lin_w || !node_terms || !xlin || !s_i || !s_j) return GDN_ERR_ARG;
  Plan pl; int threads;
  const int rc = make_plan(MODE_PROJECT, batch, n, w, d, 0, &pl, &threads);
  if (rc != GDN_OK) return rc;
  Args a = {};
  a.x = x; a.lin_w = lin_w; a.node_terms = node_terms;
  a.xlin_out = xlin; a.si_out = s_i; a.sj_out = s_j;
  return dispatch_window<MODE_PROJECT>(pl, a, threads, (hipStream_t)stream);
}

extern "C" int gdn_attn_aggregate_fwd(const float* xlin, const float* s_i, const float* s_j,
                                      const uint16_t* nbr, const int32_t* deg, const float* bias,
                                      int batch, int n, int d, int k, float* z, float* alpha,
                                      void* stream) {
  if (!xlin || !s_i || !s_j || !nbr || !deg || !bias || !z) return GDN_ERR_ARG;
  Plan pl; int threads;
  const int rc = make_plan(MODE_ATTN, batch, n, 0, d, k, &pl, &threads);
  if (rc != GDN_OK) return rc;
  Args a = {};
  a.xlin_in = xlin; a.si_in = s_i; a.sj_in = s_j; a.nbr = nbr; a.deg = deg; a.gnn_bias = bias;
  a.z = z; a.alpha = alpha;
  return dispatch_window<MODE_ATTN>(pl, a, threads, (hipStream_t)stream);
}

extern "C" int gdn_forward_fused(const float* x, const float* lin_w, const float* node_terms,
                                 const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                                 const float* emb, const float* bn1_affine, const float* bn2_affine,
                                 const float* out_w, const float* out_b, int batch, int n, int w, int d,
                                 int k, float* out, void* stream) {
  if (!x || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine ||
      !bn2_affine || !out_w || !out_b || !out)
    return GDN_ERR_ARG;
  Plan pl; int threads;
  const int rc = make_plan(MODE_FUSED, batch, n, w, d, k, &pl, &threads);
  if (rc != GDN_OK) return rc;
  Args a = {};
  a.x = x; a.lin_w = lin_w; a.node_terms = node_terms; a.nbr = nbr; a.deg = deg;
  a.gnn_bias = gnn_bias; a.emb = emb; a.bn1 = bn1_affine; a.bn2 = bn2_affine;
  a.out_w = out_w; a.out_b = out_b; a.out = out;
  return dispatch_window<MODE_FUSED>(pl, a, threads, (hipStream_t)stream);
}

extern "C" int gdn_head_fwd(const float* z, const float* emb, const float* bn1_affine,
                            const float* bn2_affine, const float* out_w, const float* out_b, int batch,
                            int n, int d, float* out, float* h2, void* stream) {
  if (!z || !emb || !bn1_affine || !bn2_affine || !out_w || !out_b || !out || batch <= 0 || n <= 0)
    return GDN_ERR_ARG;
  const int rows = batch * n;
  const int grid = min((rows + 15) / 16, gdn_cu_count() * 8);
  hipStream_t st = (hipStream_t)stream;
  switch (d) {
    case 16: hipLaunchKernelGGL(gdn_head_kernel<16>, dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 32: hipLaunchKernelGGL(gdn_head_kernel<32>, dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 64: hipLaunchKernelGGL(gdn_head_kernel<64>, dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 128: hipLaunchKernelGGL(gdn_head_kernel<128>, dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    default: return GDN_ERR_UNSUPPORTED;
  }
  return gdn_launch_status();
}
