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

#include <stdlib.h>

#ifndef GDN_GATHER_PRIO
#define GDN_GATHER_PRIO 2         // wave priority inside the gather rounds of long lists (see aggregate_target)
#endif
#ifndef GDN_GATHER_CHUNK
#define GDN_GATHER_CHUNK 8        // fused kernel: row fetches in flight per wave
#endif
// register buffers of 8 row fetches in the hand-pipelined gather: 2 = fetch(h+1) overlaps fma(h)
#ifndef GDN_PIPE_FUSED
#define GDN_PIPE_FUSED 3
#endif
#ifndef GDN_PIPE_ATTN
#define GDN_PIPE_ATTN 3
#endif
#define GDN_PIPE_DEPTH(MODE) ((MODE) == MODE_ATTN ? GDN_PIPE_ATTN : GDN_PIPE_FUSED)
#ifndef GDN_GATHER_CHUNK_ATTN
#define GDN_GATHER_CHUNK_ATTN 4   // K8 kernel also holds the next tile's 32 prefetch registers
#endif

namespace {

template <int V>
struct Pack {
  float v[V];
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += a * src over V columns, as v_pk_fma_f32 on column pairs (the SLP vectoriser does not
// pack this loop reliably; scalar v_fmac doubles the instruction count of the gather)
template <int V>
__device__ __forceinline__ void axpy_pack(float a, const Pack<V>& src, Pack<V>& acc) {
  if constexpr (V >= 2) {
    const f32x2 a2 = {a, a};
#pragma unroll
    for (int v = 0; v < V; v += 2) {
      f32x2 s2 = {src.v[v], src.v[v + 1]};
      f32x2 c2 = {acc.v[v], acc.v[v + 1]};
      c2 = __builtin_elementwise_fma(a2, s2, c2);
      acc.v[v] = c2[0];
      acc.v[v + 1] = c2[1];
    }
  } else {
    acc.v[0] = fmaf(a, src.v[0], acc.v[0]);
  }
}

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
  int wl_lds;    // 1: lin.weight staged (permuted) in LDS, 0: lane blocks loaded from global
  int mfma;      // 1: projection on v_mfma_f32_32x32x2_f32 (x tile at odd pitch xp = wpm + 1)
  int wpm, xp;   // MFMA path: padded window (16 / 32) and x-tile pitch in floats
  int dfull;     // row stride of xlin / z / emb / lin.weight rows in global memory (= model dim)
  int nslices;   // gridDim.y: column slices of width d when the full-width tile exceeds LDS
  int off_xl, off_si, off_sj, off_deg, off_wl, off_nbr, off_xs;  // float offsets
  int lds_bytes;
};

enum { MODE_PROJECT = 0, MODE_ATTN = 1, MODE_FUSED = 2 };

// first model column of this workgroup's slice (gridDim.y slices of the tile width)
#define GDN_COL0(D) ((int)blockIdx.y * (D))

struct Args {
  // inputs
  const float* x;           // [B, n, w], or the raw series [n, series_len] when series_len > 0
  int series_len;           // > 0: window b = series[:, series_first + b : series_first + b + w]
  int series_first;
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
  int x_bf16;       // bf16 storage: x holds bfloat16 bits, the projected tile is rounded to bf16
  // MODE_FUSED, optional: gate[0] = the range flag a preceding matrix-core launch may have set, gate[1] = a ticket.
  // gate[0] == 0: every workgroup returns at once (the launch costs its dispatch only); otherwise the launch
  // recomputes all windows in fp32 and the last workgroup to finish clears both words for the next call.
  int* gate;
};

// x element idx of a window whose first element is `elems` past a.x, for either storage type
__device__ __forceinline__ const float* x_base(const Args& a, size_t elems) {
  return a.x_bf16 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(a.x) + elems) : a.x + elems;
}
__device__ __forceinline__ float x_elem(const Args& a, const float* base, size_t idx) {
  return a.x_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(base)[idx] << 16) : base[idx];
}
__device__ __forceinline__ float round_to_bf16(float v) {   // round to nearest even, as torch's .bfloat16()
  const unsigned u = __float_as_uint(v);
  return __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
}

// End of a gated MODE_FUSED launch (Args::gate): every workgroup has read the flag by the time the last ticket is
// drawn, so the workgroup that draws it clears flag and ticket for the next call.
__device__ __forceinline__ void gate_release(const Args& a) {
  if (!a.gate) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = (int)(gridDim.x * gridDim.y);
    const int t = __hip_atomic_fetch_add(a.gate + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == total - 1) {
      __hip_atomic_store(a.gate + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.gate, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ projection phase
// lin.weight lives in LDS for the whole workgroup, permuted so that a lane fetches its
// [VEC x WCH] block for W-chunk wc with conflict-free ds_read_b128:
//   float4 index = ((wc * VEC + v) * (WCH/4) + c4) * (D/VEC) + lane_slot,   lane_slot = d0 / VEC
// (consecutive lanes -> consecutive 16-B slots).  Columns >= w are stored as 0.
template <int D, int WCH>
__device__ __forceinline__ void stage_weights(const Plan& pl, const Args& a, float* wlds) {
  using G = Geo<D>;
  constexpr int LS = D / G::VEC;  // lane slots
  const int total = D * pl.wp;     // floats
  for (int t = threadIdx.x; t < total; t += blockDim.x) {
    const int e = t & 3;                 // element of the float4
    const int q = t >> 2;
    const int slot = q % LS;
    const int r = q / LS;                // (wc * VEC + v) * (WCH/4) + c4
    const int c4 = r % (WCH / 4);
    const int vv = (r / (WCH / 4)) % G::VEC;
    const int wc = r / ((WCH / 4) * G::VEC);
    const int col = wc * WCH + c4 * 4 + e;
    const int drow = slot * G::VEC + vv;
    wlds[t] = col < pl.w ? a.lin_w[(size_t)(GDN_COL0(D) + drow) * pl.w + col] : 0.f;
  }
}

template <int D, int WCH>
__device__ __forceinline__ void load_lane_weights(const Plan& pl, const Args& a, const float* wlds, int wc,
                                                  float (&wl)[Geo<D>::VEC][WCH]) {
  using G = Geo<D>;
  constexpr int LS = D / G::VEC;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int slot = (grp % G::NS) * 16 + l16;
  if (!pl.wl_lds) {   // big tiles: no LDS to spare, read the block through L2
    const int d0 = slot * G::VEC;
#pragma unroll
    for (int v = 0; v < G::VEC; ++v)
#pragma unroll
      for (int c = 0; c < WCH; ++c) {
        const int col = wc * WCH + c;
        wl[v][c] = col < pl.w ? a.lin_w[(size_t)(GDN_COL0(D) + d0 + v) * pl.w + col] : 0.f;
      }
    return;
  }
#pragma unroll
  for (int v = 0; v < G::VEC; ++v)
#pragma unroll
    for (int c4 = 0; c4 < WCH / 4; ++c4) {
      const float4 t = *reinterpret_cast<const float4*>(
          wlds + ((size_t)((wc * G::VEC + v) * (WCH / 4) + c4) * LS + slot) * 4);
      wl[v][c4 * 4] = t.x; wl[v][c4 * 4 + 1] = t.y; wl[v][c4 * 4 + 2] = t.z; wl[v][c4 * 4 + 3] = t.w;
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
      if (TO_GLOBAL && last && blockIdx.y == 0) {
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
    if (a.x_bf16 && last) {
#pragma unroll
      for (int v = 0; v < G::VEC; ++v) acc.v[v] = round_to_bf16(acc.v[v]);
    }
    st_pack<G::VEC>(xl + (size_t)row * D + d0, acc);
    if (TO_GLOBAL && last)
      st_pack<G::VEC>(a.xlin_out + ((size_t)b * pl.n + row) * pl.dfull + GDN_COL0(D) + d0, acc);
  }
}

// ------------------------------------------------------------------ attention + aggregation
// acc += alpha_q * xlin[j_q] for the 16 neighbours q held by the lanes of this row.  Step S uses
// the pair rotated by S lanes; every rotation is taken from the ORIGINAL registers (row_ror:S), so
// the 16 LDS addresses are independent and all 16 ds_read_b128 can be in flight at once.
template <int D, int S, int CH>
__device__ __forceinline__ void gather_steps(const char* xl_lane, float al, int jb,
                                             Pack<Geo<D>::VEC>& acc) {
  using G = Geo<D>;
  if constexpr (S < 16) {
    float a_s;
    int j_s;
    if constexpr (S == 0) {
      a_s = al;
      j_s = jb;
    } else {
      a_s = dpp_f<0x120 + S>(al);
      j_s = dpp_i<0x120 + S>(jb);
    }
    const Pack<G::VEC> src = ld_pack<G::VEC>(reinterpret_cast<const float*>(xl_lane + j_s));
    axpy_pack<G::VEC>(a_s, src, acc);
    // keep at most CH row fetches in flight: bounds the VGPRs the scheduler may spend
    if constexpr ((S + 1) % CH == 0 && S + 1 < 16) __builtin_amdgcn_sched_barrier(0);
    gather_steps<D, S + 1, CH>(xl_lane, al, jb, acc);
  }
}

// (Measured, round 3, at the 512-sensor shape — k = 64, where this loop is VALU-issue bound: handing the (weight,
// row offset) pairs round through LDS instead of DPP — the lane group writes its 16 pairs, every lane reads pair S
// back as a broadcast — was SLOWER: 11.7 ms per 32768 windows with the pair read right before its row fetch (two
// chained LDS round trips per step), 10.4 ms with all 16 pairs preloaded by 8 ds_read_b128, against 9.1 ms for the
// rotation below.)
// Half a round (8 neighbours) at a time, software-pipelined by hand: the 8 LDS fetches of the NEXT
// half are issued before the 16 packed FMAs of the current one (the machine scheduler left to itself
// keeps only two fetches in flight).  S0 = first rotation step of the half (0 or 8).
template <int ROT>
__device__ __forceinline__ int ror_i(int v) {
  if constexpr (ROT == 0) return v;
  else return dpp_i<0x120 + ROT>(v);
}
template <int ROT>
__device__ __forceinline__ float ror_f(float v) {
  if constexpr (ROT == 0) return v;
  else return dpp_f<0x120 + ROT>(v);
}

template <int D, int S0, int N = 8, int S = 0>
__device__ __forceinline__ void fetch_half(const char* xl_lane, int jb, Pack<Geo<D>::VEC> (&buf)[N]) {
  if constexpr (S < N) {
#if defined(GDN_ABLATE) && GDN_ABLATE == 2   // diagnostic build: no LDS row fetches (results are wrong)
    {
      const int o = ror_i<S0 + S>(jb);
#pragma unroll
      for (int v = 0; v < Geo<D>::VEC; ++v) buf[S].v[v] = __int_as_float(o + v);
    }
#else
    buf[S] = ld_pack<Geo<D>::VEC>(reinterpret_cast<const float*>(xl_lane + ror_i<S0 + S>(jb)));
#endif
    fetch_half<D, S0, N, S + 1>(xl_lane, jb, buf);
  }
}

template <int D, int S0, int N = 8, int S = 0>
__device__ __forceinline__ void fma_half(float al, const Pack<Geo<D>::VEC> (&buf)[N], Pack<Geo<D>::VEC>& acc) {
  if constexpr (S < N) {
#if defined(GDN_ABLATE) && GDN_ABLATE == 1   // diagnostic build: fetches kept alive, no FMAs / alpha moves
#pragma unroll
    for (int v = 0; v < Geo<D>::VEC; ++v) asm volatile("" ::"v"(buf[S].v[v]));
#else
    axpy_pack<Geo<D>::VEC>(ror_f<S0 + S>(al), buf[S], acc);
#endif
    fma_half<D, S0, N, S + 1>(al, buf, acc);
  }
}

// quarter rounds (4 neighbours), double buffered: fetch(q+1) is issued before fma(q), through all
// MAXR rounds of the target.  Q = global quarter index 0 .. 4*MAXR-1.
template <int D, int MAXR, int Q = 0>
__device__ __forceinline__ void quarter_pipeline(const char* xl_lane, const float (&al)[MAXR], const int (&jb)[MAXR],
                                                 Pack<Geo<D>::VEC> (&b0)[4], Pack<Geo<D>::VEC> (&b1)[4],
                                                 Pack<Geo<D>::VEC>& acc) {
  if constexpr (Q < 4 * MAXR) {
    if constexpr (Q + 1 < 4 * MAXR) {
      constexpr int NR = (Q + 1) / 4, NS = ((Q + 1) % 4) * 4;
      if constexpr ((Q + 1) % 2 == 0) fetch_half<D, NS, 4>(xl_lane, jb[NR], b0);
      else fetch_half<D, NS, 4>(xl_lane, jb[NR], b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int R = Q / 4, S0 = (Q % 4) * 4;
    if constexpr (Q % 2 == 0) fma_half<D, S0, 4>(al[R], b0, acc);
    else fma_half<D, S0, 4>(al[R], b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    quarter_pipeline<D, MAXR, Q + 1>(xl_lane, al, jb, b0, b1, acc);
  }
}

struct AggCtx {
  const float* si;
  const float* sj;
  const uint16_t* degs;
  const uint16_t* nbr_lds;   // LDS copy of the neighbour lists, or null
  const uint16_t* nbr_glb;   // the lists in global memory (used when they do not fit in LDS)
  const char* xl_lane;
};

// one list entry; where the lists live is a COMPILE-TIME property of the kernel variant (a pointer
// selected at run time would degrade every access to a FLAT load)
template <bool IN_LDS>
__device__ __forceinline__ int nbr_entry(const AggCtx& c, int idx) {
  if constexpr (IN_LDS) return (int)c.nbr_lds[idx];
  else return (int)c.nbr_glb[idx];
}

// Softmax weights + weighted neighbour sum of ONE target row.
//   MAXR = 1, 2 : the list pitch is exactly 16*MAXR; every round runs unconditionally (padding
//                 slots hold the sentinel, weight 0), logits in registers, straight-line code;
//   MAXR = 5    : up to 5 rounds (wave-uniform skip of the unused ones), logits in registers;
//   MAXR = 0    : any length, logits recomputed per pass.
// Returns the aggregate (without bias) in `acc`.
template <int D, int MODE, int MAXR, bool NL>
__device__ __forceinline__ void aggregate_target(const Plan& pl, const Args& a, const AggCtx& c, int b, int i,
                                                 int slice, int l16, Pack<Geo<D>::VEC>& acc) {
  using G = Geo<D>;
  const float sti = c.si[i];
  const int row0 = i * pl.pitch;
  float* arow = nullptr;
  if constexpr (MODE == MODE_ATTN) {
    if (a.alpha && slice == 0 && blockIdx.y == 0) arow = a.alpha + ((size_t)b * pl.n + i) * pl.pitch;
  }
#pragma unroll
  for (int v = 0; v < G::VEC; ++v) acc.v[v] = 0.f;

  if constexpr (MAXR > 0) {
    // rounds in use.  MAXR = 5: from the target's own degree — the pitch leaves room for k + 1 entries, but a
    // sensor is (almost always) in its own top-k, so k = 64 fills exactly 4 rounds and the fifth holds sentinels
    // only: skipping it is 20 % of the gather work of BASELINE configs[4].  (The four targets of a wave may in
    // principle differ: the branch is then taken per 16-lane group.)
    const int nr = MAXR <= 2 ? MAXR : min(pl.pitch >> 4, ((int)c.degs[i] + 15) >> 4);
    int jn[MAXR];
    float e[MAXR];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) jn[r] = r < nr ? nbr_entry<NL>(c, row0 + r * 16 + l16) : pl.n;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      e[r] = leaky(sti + c.sj[jn[r]]);   // sentinel: s_j[n] = -inf
      m = fmaxf(m, e[r]);
    }
    m = row16_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      e[r] = __expf(e[r] - m);   // exp(-inf) = 0 in the padding slots
      sum += e[r];
    }
    sum = row16_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum + GDN_SOFTMAX_EPS);
    if constexpr (MAXR <= 2) {
      // straight-line, hand-pipelined: fetch(h+1) is issued before fma(h)
      float al[MAXR];
      int jb[MAXR];
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        al[r] = e[r] * inv;
        jb[r] = jn[r] * (D * 4);
        if constexpr (MODE == MODE_ATTN) {
          if (arow) arow[r * 16 + l16] = al[r];
        }
      }
      if constexpr (GDN_PIPE_DEPTH(MODE) == 2) {
        Pack<G::VEC> b0[8], b1[8];
        fetch_half<D, 0>(c.xl_lane, jb[0], b0);
        __builtin_amdgcn_sched_barrier(0);
        fetch_half<D, 8>(c.xl_lane, jb[0], b1);
        __builtin_amdgcn_sched_barrier(0);
        fma_half<D, 0>(al[0], b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MAXR == 2) {
          fetch_half<D, 0>(c.xl_lane, jb[1], b0);
          __builtin_amdgcn_sched_barrier(0);
        }
        fma_half<D, 8>(al[0], b1, acc);
        if constexpr (MAXR == 2) {
          __builtin_amdgcn_sched_barrier(0);
          fetch_half<D, 8>(c.xl_lane, jb[1], b1);
          __builtin_amdgcn_sched_barrier(0);
          fma_half<D, 0>(al[1], b0, acc);
          __builtin_amdgcn_sched_barrier(0);
          fma_half<D, 8>(al[1], b1, acc);
        }
      } else if constexpr (GDN_PIPE_DEPTH(MODE) == 3) {
        Pack<G::VEC> q0[4], q1[4];
        fetch_half<D, 0, 4>(c.xl_lane, jb[0], q0);
        __builtin_amdgcn_sched_barrier(0);
        quarter_pipeline<D, MAXR>(c.xl_lane, al, jb, q0, q1, acc);
      } else {
        // one register buffer: 8 fetches in flight, then their FMAs
        Pack<G::VEC> b0[8];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
          fetch_half<D, 0>(c.xl_lane, jb[r], b0);
          __builtin_amdgcn_sched_barrier(0);
          fma_half<D, 0>(al[r], b0, acc);
          __builtin_amdgcn_sched_barrier(0);
          fetch_half<D, 8>(c.xl_lane, jb[r], b0);
          __builtin_amdgcn_sched_barrier(0);
          fma_half<D, 8>(al[r], b0, acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // the gather rounds ahead of the softmax / epilogue arithmetic of the other waves on the SIMD (their LDS
      // fetches are what the loop waits for): configs[4], 4096 windows, same box: 1124 -> 1075 us (bf16-stored
      // windows 1100 -> 1055), priorities 1, 2 and 3 alike
      __builtin_amdgcn_s_setprio(GDN_GATHER_PRIO);
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        if (r < nr) {
          const float al = e[r] * inv;
          if constexpr (MODE == MODE_ATTN) {
            if (arow) arow[r * 16 + l16] = al;
          }
          gather_steps<D, 0, (MODE == MODE_ATTN ? GDN_GATHER_CHUNK_ATTN : GDN_GATHER_CHUNK)>(c.xl_lane, al, jn[r] * (D * 4), acc);
        } else if constexpr (MODE == MODE_ATTN) {
          if (arow && r * 16 < pl.pitch) arow[r * 16 + l16] = 0.f;      // a round of sentinels: weights 0
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    const int rounds = (c.degs[i] + 15) >> 4;
    float m = -INFINITY;
    for (int r = 0; r < rounds; ++r)
      m = fmaxf(m, leaky(sti + c.sj[nbr_entry<NL>(c, row0 + r * 16 + l16)]));
    m = row16_max(m);
    float sum = 0.f;
    for (int r = 0; r < rounds; ++r)
      sum += __expf(leaky(sti + c.sj[nbr_entry<NL>(c, row0 + r * 16 + l16)]) - m);
    sum = row16_sum(sum);
    const float inv = __builtin_amdgcn_rcpf(sum + GDN_SOFTMAX_EPS);
    for (int r = 0; r < rounds; ++r) {
      const int p = r * 16 + l16;
      const int j = nbr_entry<NL>(c, row0 + p);
      const float al = __expf(leaky(sti + c.sj[j]) - m) * inv;
      if constexpr (MODE == MODE_ATTN) {
        if (arow) arow[p] = al;
      }
      gather_steps<D, 0, (MODE == MODE_ATTN ? GDN_GATHER_CHUNK_ATTN : GDN_GATHER_CHUNK)>(c.xl_lane, al, j * (D * 4), acc);
    }
    if constexpr (MODE == MODE_ATTN) {
      // zero the alpha slots of rounds this target does not use
      if (arow)
        for (int p = rounds * 16 + l16; p < pl.pitch; p += 16) arow[p] = 0.f;
    }
  }
}

// LST encodes the neighbour-list variant of the kernel: 1, 2, 5 = lists in LDS with exactly 1 / exactly
// 2 / up to 5 rounds of 16 (logits in registers); 6 = up to 5 rounds, lists read from global (they do
// not fit beside a big tile); 0 = any length, lists in global, logits recomputed per pass.
template <int D, int MODE, int LST>
__device__ __forceinline__ void aggregate_window(const Plan& pl, const Args& a, float* smem, int b) {
  using G = Geo<D>;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const int slot = grp / G::NS, slice = grp % G::NS;
  const int tpp = (blockDim.x >> 4) / G::NS;
  const int d0 = slice * 64 + l16 * G::VEC;
  AggCtx c;
  c.si = smem + pl.off_si;
  c.sj = smem + pl.off_sj;
  c.degs = reinterpret_cast<const uint16_t*>(smem + pl.off_deg);
  c.nbr_lds = reinterpret_cast<const uint16_t*>(smem + pl.off_nbr);
  c.nbr_glb = a.nbr;
  c.xl_lane = reinterpret_cast<const char*>(smem + pl.off_xl + d0);

  // per-lane constants of the epilogue
  const int gcol = GDN_COL0(D) + d0;   // this lane's first MODEL column
  const Pack<G::VEC> bias = ld_pack<G::VEC>(a.gnn_bias + gcol);
  Pack<G::VEC> sc1, sh1, sc2, sh2, wo;
  float out_b = 0.f;
  if constexpr (MODE == MODE_FUSED) {
    sc1 = ld_pack<G::VEC>(a.bn1 + gcol);
    sh1 = ld_pack<G::VEC>(a.bn1 + pl.dfull + gcol);
    sc2 = ld_pack<G::VEC>(a.bn2 + gcol);
    sh2 = ld_pack<G::VEC>(a.bn2 + pl.dfull + gcol);
    wo = ld_pack<G::VEC>(a.out_w + gcol);
    out_b = blockIdx.y == 0 ? a.out_b[0] : 0.f;
  }

  // (Measured, round 3: fetching the NEXT target's list entries and embedding row one target ahead at the 512-sensor
  // shape changed nothing — 9.13 vs 9.34 ms per 32768 windows: the kernel is VALU-issue bound there,
  // SQ_WAIT_INST_ANY 5 % of the wave cycles, profiles/r03_sq_counters_config4_B4096.json.)
  for (int i = slot; i < pl.n; i += tpp) {
    Pack<G::VEC> emb_i;
    if constexpr (MODE == MODE_FUSED) emb_i = ld_pack<G::VEC>(a.emb + (size_t)i * pl.dfull + gcol);
    Pack<G::VEC> acc;
    aggregate_target<D, MODE, (LST == 6 ? 5 : LST), (LST >= 1 && LST <= 5)>(pl, a, c, b, i, slice, l16, acc);
#pragma unroll
    for (int v = 0; v < G::VEC; ++v) acc.v[v] += bias.v[v];
    if constexpr (MODE == MODE_ATTN) {
      st_pack<G::VEC>(a.z + ((size_t)b * pl.n + i) * pl.dfull + gcol, acc);
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
      if (l16 == 0 && slice == 0) {
        if (pl.nslices == 1) a.out[(size_t)b * pl.n + i] = part + out_b;
        else atomicAdd(&a.out[(size_t)b * pl.n + i], part + out_b);   // out zeroed by the launcher
      }
    }
  }
}

// neighbour-list length class (kernel template parameter MAXR): exactly 1 or 2 rounds of 16
// (k <= 15 / k <= 31), up to 5 rounds (k <= 79), or 0 = any length (logits recomputed per pass).
// ------------------------------------------------------------------ projection on the matrix cores
// xlin[32-row block, 32-col block] = sum_k x[row, k] * lin[col, k] with v_mfma_f32_32x32x2_f32
// (exact fp32, one k-pair per instruction).  A operand: lane l holds x[row = l&31][k = 2kk + (l>>5)]
// (x tile at the odd pitch xp -> conflict-free ds_read_b32); B operand: lin[col = l&31][k], kept in
// registers for the whole workgroup; C/D: reg r of lane l = xlin[(r&3) + 8(r>>2) + 4(l>>5)][l&31].
// One wave per 32-row block.  The attention scalars s_i/s_j are one thread per sensor.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int XU>
struct XFlat {
  float v[XU];
};

// Element t of a window is src[(t / w) * row_stride + t % w]: row_stride = w for materialised
// windows x[b] (src = x + b*n*w), row_stride = series_len when the window is read straight from the
// raw [n, series_len] series (src = series + first + b; reference datasets/TimeDataset.py:46-49
// builds x[b] = data[:, b : b+w] on the host, a w-fold redundant copy — neighbouring windows share
// all but one column, so the series is fetched from HBM once).  One code path, no branch.
template <int XU>
__device__ __forceinline__ void xflat_load(const Args& a, const float* src, int row_stride, int cnt, int w,
                                           float inv_w, XFlat<XU>& r) {
#pragma unroll
  for (int u = 0; u < XU; ++u) {   // unconditional, clamped: stays in registers
    const int t = min((int)threadIdx.x + u * (int)blockDim.x, cnt - 1);
    const int row = (int)((t + 0.5f) * inv_w);   // exact for t < 2^16, w <= 64
    // 32-bit, window-invariant offset from a wave-uniform base: one VGPR per element
    r.v[u] = x_elem(a, src, (unsigned int)(row * row_stride + (t - row * w)));
  }
}

template <int XU>
__device__ __forceinline__ void xflat_store(float* xs, int cnt, int w, float inv_w, int xp,
                                            const XFlat<XU>& r) {
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int t = threadIdx.x + u * blockDim.x;
    if (t < cnt) {
      const int row = (int)((t + 0.5f) * inv_w);   // exact for t < 2^16, w <= 64
      xs[row * xp + (t - row * w)] = r.v[u];
    }
  }
}

// Rows [r0, r1) of the window (the x tile in LDS holds exactly these rows, row r at xs[(r - r0) * XP]): the whole
// window for the variants that stage it at once, one row chunk for the chunked variant (large n).
template <int D, int MODE, int WPM>
__device__ __forceinline__ void project_mfma(const Plan& pl, const Args& a, float* smem, int b,
                                             const float (&wb)[D / 32][WPM / 2], int r0 = 0, int r1 = -1) {
  if (r1 < 0) r1 = pl.n;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6, nw = nth >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const float* xs = smem + pl.off_xs;
  float* xl = smem + pl.off_xl;
  float* si = smem + pl.off_si;
  float* sj = smem + pl.off_sj;
  constexpr int XP = WPM + 1;
  // attention scalars: s = x_row . a + c[sensor]
  for (int t = r0 + tid; t < r1; t += nth) {
    float pi = a.node_terms[2 * GDN_A_PITCH + t];
    float pj = a.node_terms[2 * GDN_A_PITCH + pl.n + t];
    const float* xr = xs + (t - r0) * XP;
#pragma unroll
    for (int k = 0; k < WPM; ++k) {
      const float xv = xr[k];
      pi = fmaf(xv, a.node_terms[k], pi);
      pj = fmaf(xv, a.node_terms[GDN_A_PITCH + k], pj);
    }
    si[t] = pi;
    sj[t] = pj;
    if constexpr (MODE == MODE_PROJECT) {
      if (blockIdx.y == 0) {
        a.si_out[(size_t)b * pl.n + t] = pi;
        a.sj_out[(size_t)b * pl.n + t] = pj;
      }
    }
  }
  // one wave per 32-row block (its A operand is loaded once and reused for every column block).
  // Measured alternative — dealing single 32x32 tiles round-robin so 8-wave workgroups stay busy —
  // was slower at n = 127 (A reloaded per tile): 48 vs 55 Mwin/s.
  const int nrb = (r1 + 31) >> 5;
  for (int rb = (r0 >> 5) + wv; rb < nrb; rb += nw) {          // (r0 is a multiple of 32)
    const float* arow = xs + (min(rb * 32 + l32, r1 - 1) - r0) * XP + h;
    float av[WPM / 2];
#pragma unroll
    for (int kk = 0; kk < WPM / 2; ++kk) av[kk] = arow[2 * kk];
#pragma unroll
    for (int cb = 0; cb < D / 32; ++cb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < WPM / 2; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], wb[cb][kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < r1) {
          xl[row * D + cb * 32 + l32] = a.x_bf16 ? round_to_bf16(acc[r]) : acc[r];
          if constexpr (MODE == MODE_PROJECT)
            a.xlin_out[((size_t)b * pl.n + row) * pl.dfull + GDN_COL0(D) + cb * 32 + l32] = acc[r];
        }
      }
    }
  }
}

template <int D, int MODE, int WPM, int XU, int LST>
__device__ __forceinline__ void window_loop_mfma(const Plan& pl, const Args& a, float* smem) {
  const int tid = threadIdx.x, nth = blockDim.x;
  const int l32 = tid & 31, h = (tid >> 5) & 1;
  constexpr int XP = WPM + 1;
  float* xs = smem + pl.off_xs;
  // zero the x tile once: pad columns w..WPM-1 are never written again
  for (int t = tid; t < pl.n * XP; t += nth) xs[t] = 0.f;
  float wb[D / 32][WPM / 2];
#pragma unroll
  for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
    for (int kk = 0; kk < WPM / 2; ++kk) {
      const int k = 2 * kk + h;
      wb[cb][kk] = k < pl.w ? a.lin_w[(size_t)(GDN_COL0(D) + cb * 32 + l32) * pl.w + k] : 0.f;
    }
  const int cnt = pl.n * pl.w;
  const float inv_w = 1.0f / (float)pl.w;
  XFlat<XU> xr;
  const int row_stride = a.series_len > 0 ? a.series_len : pl.w;
  const size_t win_stride = a.series_len > 0 ? 1 : (size_t)cnt;   // distance between windows b, b+1
  const size_t first = a.series_len > 0 ? a.series_first : 0;
  auto load_window = [&](int bb) {
    xflat_load<XU>(a, x_base(a, first + (size_t)bb * win_stride), row_stride, cnt, pl.w, inv_w, xr);
  };
  load_window(blockIdx.x);
  __syncthreads();   // zero fill done before the first store
  xflat_store<XU>(xs, cnt, pl.w, inv_w, XP, xr);
  __syncthreads();
  // two barriers per window: [loads of b+1 | s + MFMA of b] B [aggregate b | store x of b+1] C
  for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
    const int nb = min(b + (int)gridDim.x, pl.batch - 1);   // last round: harmless re-read
    load_window(nb);                                        // lands under the math
    project_mfma<D, MODE, WPM>(pl, a, smem, b, wb);
    __syncthreads();                                        // B: tile + scalars ready, x tile free
    if constexpr (MODE == MODE_FUSED) aggregate_window<D, MODE, LST>(pl, a, smem, b);
    xflat_store<XU>(xs, cnt, pl.w, inv_w, XP, xr);
    __syncthreads();                                        // C: tile free, next x tile visible
  }
}

// Large n (the 512-sensor stress shape): the projected tile alone takes 131 KB of LDS, the x tile of a whole window
// (n x 33 floats = 68 KB) no longer fits beside it, and the VALU projection this kernel then fell back to was 60 %
// of the fused launch (1.70 of 2.35 ms per 4096 windows, profiles/r03_kernels_config4_B4096_kernel_stats.csv).
// Chunked form: the window's x values live in REGISTERS (XU per thread; a second set prefetches the next window),
// the LDS x tile holds pl.xrows rows at a time (a multiple of 32: 160 rows = 21 KB at n = 512) and the matrix-core
// projection runs chunk by chunk: [store chunk | barrier | s_i, s_j + MFMA of the chunk | barrier].
template <int XU>
__device__ __forceinline__ void xflat_store_rows(float* xs, int cnt, int w, float inv_w, int xp, int r0, int r1,
                                                 const XFlat<XU>& r) {
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int t = threadIdx.x + u * blockDim.x;
    const int row = (int)((t + 0.5f) * inv_w);   // exact for t < 2^16, w <= 64
    if (t < cnt && row >= r0 && row < r1) xs[(row - r0) * xp + (t - row * w)] = r.v[u];
  }
}

template <int D, int MODE, int WPM, int XU, int LST>
__device__ __forceinline__ void window_loop_mfma_chunked(const Plan& pl, const Args& a, float* smem) {
  const int tid = threadIdx.x, nth = blockDim.x;
  const int l32 = tid & 31, h = (tid >> 5) & 1;
  constexpr int XP = WPM + 1;
  float* xs = smem + pl.off_xs;
  const int rc = pl.xrows;                       // rows per chunk, a multiple of 32
  for (int t = tid; t < rc * XP; t += nth) xs[t] = 0.f;     // pad columns w..WPM-1 are never written again
  float wb[D / 32][WPM / 2];
#pragma unroll
  for (int cb = 0; cb < D / 32; ++cb)
#pragma unroll
    for (int kk = 0; kk < WPM / 2; ++kk) {
      const int k = 2 * kk + h;
      wb[cb][kk] = k < pl.w ? a.lin_w[(size_t)(GDN_COL0(D) + cb * 32 + l32) * pl.w + k] : 0.f;
    }
  // (Measured, round 3, 16-wave form: reloading these 32 values per window instead of keeping them live through the
  // gather loop — 128 registers a lane there — made it slower, 2.58 vs 3.81 M windows/s: more spills, not fewer.)
  const int cnt = pl.n * pl.w;
  const float inv_w = 1.0f / (float)pl.w;
  XFlat<XU> xcur, xnext;
  const int row_stride = a.series_len > 0 ? a.series_len : pl.w;
  const size_t win_stride = a.series_len > 0 ? 1 : (size_t)cnt;
  const size_t first = a.series_len > 0 ? a.series_first : 0;
  xflat_load<XU>(a, x_base(a, first + (size_t)blockIdx.x * win_stride), row_stride, cnt, pl.w, inv_w, xnext);
  __syncthreads();   // zero fill done before the first store
  for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
    xcur = xnext;
    const int nb = min(b + (int)gridDim.x, pl.batch - 1);   // last round: harmless re-read
    xflat_load<XU>(a, x_base(a, first + (size_t)nb * win_stride), row_stride, cnt, pl.w, inv_w, xnext);   // lands under the math
    for (int r0 = 0; r0 < pl.n; r0 += rc) {
      const int r1 = min(pl.n, r0 + rc);
      xflat_store_rows<XU>(xs, cnt, pl.w, inv_w, XP, r0, r1, xcur);
      __syncthreads();                                      // chunk visible
      project_mfma<D, MODE, WPM>(pl, a, smem, b, wb, r0, r1);
      __syncthreads();                                      // chunk consumed (and, after the last one, the tile is ready)
    }
    if constexpr (MODE == MODE_FUSED) {
      aggregate_window<D, MODE, LST>(pl, a, smem, b);
      __syncthreads();                                      // tile free for the next window's projection
    }
  }
}

// ------------------------------------------------------------------ the kernel
// x staging, split into "issue the global loads" and "store to LDS" (async-STAGE split): the
// loads of window b+1 are issued before window b's projection/aggregation and land under them.
// Lane (t & (wp-1)) owns column c of a row, so no division is needed; columns >= w are stored as 0.
// Fast form: wp <= 16 and at most 8 rows per thread per window (n <= 8 * blockDim/wp).
struct XRegs {
  float v[8];
};

__device__ __forceinline__ void stage_x_load(const Plan& pl, const Args& a, const float* xg, XRegs& xr) {
  const int c = threadIdx.x & (pl.wp - 1);
  const int rstep = blockDim.x / pl.wp;
  const int r = threadIdx.x / pl.wp;
  const bool live = c < pl.w;
  const int cc = live ? c : 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int rr = min(r + u * rstep, pl.n - 1);
    const float t = x_elem(a, xg, (size_t)rr * pl.w + cc);   // unconditional (clamped) load: stays in registers
    xr.v[u] = live ? t : 0.f;
  }
}

__device__ __forceinline__ void stage_x_store(const Plan& pl, float* xs, const XRegs& xr) {
  const int c = threadIdx.x & (pl.wp - 1);
  const int rstep = blockDim.x / pl.wp;
  const int r = threadIdx.x / pl.wp;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int rr = r + u * rstep;
    if (rr < pl.n) xs[rr * pl.wp + c] = xr.v[u];
  }
}

// General form: rows [r0, r1), any wp.
__device__ __forceinline__ void stage_x(const Plan& pl, const Args& a, const float* xg, float* xs, int r0, int r1) {
  const int cnt = (r1 - r0) * pl.wp;
  for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
    const int r = t / pl.wp, c = t - r * pl.wp;
    xs[t] = c < pl.w ? x_elem(a, xg, (size_t)(r0 + r) * pl.w + c) : 0.f;
  }
}

// PROJ selects the projection variant at COMPILE time (a runtime branch would make every variant
// pay the registers of the hungriest one): 0 VALU w<=8, 1 VALU w-chunks of 16,
// 2 MFMA w<=16 (<=8 x values per thread), 3 MFMA w<=16 (<=16 per thread), 4 MFMA w<=32.
template <int D, int MODE, int NT, int PROJ, int LST>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((NT == 256 ? 3 : (NT == 1024 ? 4 : 2)), (NT == 256 ? 3 : (NT == 1024 ? 4 : 2))))) void gdn_window_kernel(const Plan pl, const Args a) {
  using G = Geo<D>;
  constexpr int WCH = PROJ == 0 ? 8 : 16;
  extern __shared__ float4 smem_f4[];
  float* smem = reinterpret_cast<float*>(smem_f4);
  const int tid = threadIdx.x, nth = blockDim.x;
  if constexpr (MODE == MODE_FUSED) {
    if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // wave uniform
  }

  // once per workgroup: neighbour lists + degrees into LDS; sentinel row n of the tile = 0 and
  // s_j[n] = -inf (the padding slots of every neighbour list point there)
  if constexpr (MODE != MODE_PROJECT) {
    uint16_t* degs = reinterpret_cast<uint16_t*>(smem + pl.off_deg);
    for (int t = tid; t < pl.n; t += nth) degs[t] = (uint16_t)a.deg[t];
    for (int t = tid; t < D; t += nth) smem[pl.off_xl + pl.n * D + t] = 0.f;
    if (tid == 0) smem[pl.off_sj + pl.n] = -INFINITY;
    if constexpr (LST >= 1 && LST <= 5) {
      // n*pitch u16 = n*pitch/8 uint4 (pitch is a multiple of 16)
      const uint4* src = reinterpret_cast<const uint4*>(a.nbr);
      uint4* dst = reinterpret_cast<uint4*>(smem + pl.off_nbr);
      const int nvec = pl.n * pl.pitch / 8;
      for (int t = tid; t < nvec; t += nth) dst[t] = src[t];
    }
  }

  if constexpr (MODE == MODE_ATTN) {
    // Tile staging is split (async-STAGE): the global loads of window b+1 are issued BEFORE window
    // b's aggregation and stored to LDS after it, so HBM latency hides under the math even when
    // all workgroups of a CU run in lockstep.  Fast form: n*D/4 float4s <= 8 per thread.
    float4* tile4 = reinterpret_cast<float4*>(smem + pl.off_xl);
    float* si = smem + pl.off_si;
    float* sj = smem + pl.off_sj;
    const int nvec = pl.n * D / 4;
    const bool fast = nvec <= 8 * nth && pl.n <= nth;
    if (fast) {
      // eight named registers (an array here is demoted to scratch by the compiler); loads are
      // unconditional with a clamped index, only the LDS store is predicated
      float4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7;
      float psi, psj;
      const int tn = min(tid, pl.n - 1);
#define GDN_PRE_LOAD(u)                                                   \
  {                                                                       \
    const int tt = min(tid + u * nth, nvec - 1);                          \
    pre##u = src[(size_t)(tt / (D / 4)) * (pl.dfull / 4) + tt % (D / 4)]; \
  }
#define GDN_PRE_STORE(u)              \
  {                                   \
    const int t = tid + u * nth;      \
    if (t < nvec) tile4[t] = pre##u;  \
  }
#define GDN_PRE_ALL(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
      {
        const float4* src =
            reinterpret_cast<const float4*>(a.xlin_in + (size_t)blockIdx.x * pl.n * pl.dfull + GDN_COL0(D));
        GDN_PRE_ALL(GDN_PRE_LOAD)
        psi = a.si_in[(size_t)blockIdx.x * pl.n + tn];
        psj = a.sj_in[(size_t)blockIdx.x * pl.n + tn];
      }
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        GDN_PRE_ALL(GDN_PRE_STORE)
        if (tid < pl.n) {
          si[tid] = psi;
          sj[tid] = psj;
        }
        __syncthreads();
        const int nb = min(b + (int)gridDim.x, pl.batch - 1);   // last round re-reads its own tile
        {
          const float4* src =
              reinterpret_cast<const float4*>(a.xlin_in + (size_t)nb * pl.n * pl.dfull + GDN_COL0(D));
          GDN_PRE_ALL(GDN_PRE_LOAD)
          psi = a.si_in[(size_t)nb * pl.n + tn];
          psj = a.sj_in[(size_t)nb * pl.n + tn];
        }
        aggregate_window<D, MODE, LST>(pl, a, smem, b);
        __syncthreads();  // the tile is overwritten by the next window
      }
#undef GDN_PRE_ALL
#undef GDN_PRE_STORE
#undef GDN_PRE_LOAD
    } else {
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        const float4* src =
            reinterpret_cast<const float4*>(a.xlin_in + (size_t)b * pl.n * pl.dfull + GDN_COL0(D));
        for (int t = tid; t < nvec; t += nth)
          tile4[t] = src[(size_t)(t / (D / 4)) * (pl.dfull / 4) + t % (D / 4)];
        for (int t = tid; t < pl.n; t += nth) {
          si[t] = a.si_in[(size_t)b * pl.n + t];
          sj[t] = a.sj_in[(size_t)b * pl.n + t];
        }
        __syncthreads();
        aggregate_window<D, MODE, LST>(pl, a, smem, b);
        __syncthreads();
      }
    }
  } else {
    if constexpr (PROJ == 5) {
      static_assert(D >= 32, "MFMA projection needs d >= 32");
      window_loop_mfma_chunked<D, MODE, 32, (NT == 1024 ? 16 : 32), LST>(pl, a, smem);
      if constexpr (MODE == MODE_FUSED) gate_release(a);
      return;
    } else if constexpr (PROJ >= 2) {
      static_assert(D >= 32 || PROJ < 2, "MFMA projection needs d >= 32");
      window_loop_mfma<D, MODE, (PROJ == 4 ? 32 : 16), (PROJ == 2 ? 8 : 16), LST>(pl, a, smem);
      if constexpr (MODE == MODE_FUSED) gate_release(a);
      return;
    }
    float* xs = smem + pl.off_xs;
    float* wlds = smem + pl.off_wl;
    const int nch = pl.wp / WCH;
    if (pl.wl_lds) stage_weights<D, WCH>(pl, a, wlds);
    // (visibility of wlds: the first __syncthreads() below precedes its first read)
    const bool fast = nch == 1 && pl.xrows >= pl.n && pl.n <= 8 * (int)(blockDim.x / pl.wp);
    if (fast) {
      XRegs xr;
      stage_x_load(pl, a, x_base(a, (size_t)blockIdx.x * pl.n * pl.w), xr);
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        stage_x_store(pl, xs, xr);
        __syncthreads();
        const int nb = min(b + (int)gridDim.x, pl.batch - 1);   // last round: harmless re-read
        stage_x_load(pl, a, x_base(a, (size_t)nb * pl.n * pl.w), xr);   // lands under the math
        {
          float wl[G::VEC][WCH];
          load_lane_weights<D, WCH>(pl, a, wlds, 0, wl);
          project_chunk<D, WCH, MODE == MODE_PROJECT>(pl, a, smem, b, 0, pl.n, 0, 1, wl);
        }
        __syncthreads();
        if constexpr (MODE == MODE_FUSED) {
          aggregate_window<D, MODE, LST>(pl, a, smem, b);
          __syncthreads();
        }
      }
    } else {
      for (int b = blockIdx.x; b < pl.batch; b += gridDim.x) {
        for (int r0 = 0; r0 < pl.n; r0 += pl.xrows) {
          const int r1 = min(pl.n, r0 + pl.xrows);
          stage_x(pl, a, x_base(a, (size_t)b * pl.n * pl.w), xs, r0, r1);
          __syncthreads();
          for (int wc = 0; wc < nch; ++wc) {
            // the weight block is re-read from LDS every time instead of living in 64 VGPRs
            // across the aggregation phase
            float wl[G::VEC][WCH];
            load_lane_weights<D, WCH>(pl, a, wlds, wc, wl);
            project_chunk<D, WCH, MODE == MODE_PROJECT>(pl, a, smem, b, r0, r1, wc, nch, wl);
          }
          __syncthreads();
        }
        if constexpr (MODE == MODE_FUSED) {
          aggregate_window<D, MODE, LST>(pl, a, smem, b);
          __syncthreads();
        }
      }
    }
  }
  if constexpr (MODE == MODE_FUSED) gate_release(a);
}

// ------------------------------------------------------------------ head (staged eval path)
// z[BN, d] -> out[BN]; 16 lanes per row (VEC columns each; d = 128: two passes per lane).
__device__ __forceinline__ float head_load(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float head_load(const uint16_t* p, size_t i) { return __uint_as_float((unsigned)p[i] << 16); }

template <int D, typename ZT>
__global__ __launch_bounds__(256) void gdn_head_kernel(const ZT* __restrict__ z,
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
  constexpr int U = 4;   // rows in flight per 16-lane row group (memory-level parallelism)
  for (int row0 = (blockIdx.x * rpb + (threadIdx.x >> 4)) * U; row0 < rows; row0 += gridDim.x * rpb * U) {
    float zv[U][CPL], ev[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = min(row0 + u, rows - 1);   // clamped: loads stay unconditional
      const int s = row % n;
#pragma unroll
      for (int v = 0; v < CPL; ++v) {
        zv[u][v] = head_load(z, (size_t)row * D + d0 + v);
        ev[u][v] = emb[(size_t)s * D + d0 + v];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = row0 + u;
      float part = 0.f;
#pragma unroll
      for (int v = 0; v < CPL; ++v) {
        float h = fmaxf(fmaf(zv[u][v], sc1[v], sh1[v]), 0.f);
        h *= ev[u][v];
        h = fmaxf(fmaf(h, sc2[v], sh2[v]), 0.f);
        if (h2 && row < rows) h2[(size_t)row * D + d0 + v] = h;
        part = fmaf(h, wo[v], part);
      }
      part = row16_sum(part);
      if (l16 == 0 && row < rows) out[row] = part + ob;
    }
  }
}

// ------------------------------------------------------------------ host side
int make_plan(int mode, int batch, int n, int w, int d, int k, Plan* pl, int* threads) {
  if (batch <= 0 || n <= 0 || d <= 0) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  if (n > 4096) return GDN_ERR_UNSUPPORTED;
  pl->n = n; pl->d = d; pl->w = w; pl->k = k; pl->batch = batch;
  pl->dfull = d; pl->nslices = 1;
  if (d == 128 && (n + 1) * d * 4 + 4 * (n + 8) * 4 > 150 * 1024) {
    // the full-width tile does not fit LDS: two workgroups per window, 64 columns each (the
    // attention scalars do not depend on the slice; the head sums the slices' partial outputs)
    d = 64;
    pl->d = 64;
    pl->nslices = 2;
  }
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
  const int npad = (n + 1 + 3) & ~3;   // +1: sentinel entry / row
  int off = 0;
  pl->off_xl = off; off += (n + 1) * d;
  pl->off_si = off; off += npad;
  pl->off_sj = off; off += npad;
  pl->off_deg = off; off += (npad / 2 + 3) & ~3;
  pl->off_wl = off;
  pl->wl_lds = 0; pl->mfma = 0; pl->wpm = 0; pl->xp = 0;
  const int nbr_bytes = n * pl->pitch * 2;
  const int base_bytes = off * 4;
  if (base_bytes > LDS_MAX) return GDN_ERR_UNSUPPORTED;
  // threads: small tiles run 256-thread workgroups (several per CU), big tiles own the CU
  const int est = base_bytes + (mode != MODE_PROJECT ? nbr_bytes : 0) + n * (pl->wp + 1) * 4;
  *threads = est > 80 * 1024 ? 512 : 256;
  {   // tuning knob GDN_THREADS = 256 or 512 (read once per process)
    const int v = GDN_ENV_INT_ONCE("GDN_THREADS", 0);
    if (v == 256 || v == 512) *threads = v;
  }
  // projection on the matrix cores: d >= 32, w <= 32, whole window staged at once
  if (mode != MODE_ATTN && d >= 32 && w <= 32 && !GDN_ENV_INT_ONCE("GDN_NO_MFMA", 0)) {
    const int wpm = w <= 16 ? 16 : 32;
    const int xs_bytes = n * (wpm + 1) * 4;
    const int need = base_bytes + xs_bytes;
    if (need <= LDS_MAX && n * w <= 16 * *threads) {
      pl->mfma = 1; pl->wpm = wpm; pl->xp = wpm + 1;
    } else if (*threads == 512 && n * w <= 32 * 512) {
      // chunked form (window_loop_mfma_chunked): x in registers, the LDS x tile holds a row chunk — as many 32-row
      // blocks as fit beside the projected tile (512 sensors at d = 64: 160 rows)
      const int rows = ((LDS_MAX - base_bytes) / (33 * 4)) & ~31;
      if (rows >= 32) {
        pl->mfma = 2; pl->wpm = 32; pl->xp = 33;
        pl->xrows = rows < ((n + 31) & ~31) ? rows : ((n + 31) & ~31);
        // 16 waves per workgroup (4 per SIMD at 128 registers) when the window's x values fit 16 per thread: the
        // one workgroup a CU holds at this tile size is latency bound with 8.  configs[4], 32768 windows: 3.51 ->
        // 3.81 M windows/s (bf16-stored windows 3.14 -> 3.63 M).  Only the fused forward of launches that give a
        // workgroup several windows: 512-window launches were 7 % slower with 16 waves, and the staged projection
        // of a training step (157 spilled registers at 128) 5 %.  GDN_BIG_THREADS=512 keeps 8 waves (A/B runs).
        if (mode == MODE_FUSED && n * w <= 16 * 1024 && batch >= 4 * gdn_cu_count() &&
            GDN_ENV_INT_ONCE("GDN_BIG_THREADS", 1024) == 1024)
          *threads = 1024;
      }
    }
  }
  if (!pl->mfma && mode != MODE_ATTN &&
      (off + d * pl->wp) * 4 + 16 * pl->wp * 4 + 2048 <= LDS_MAX / 2) {
    pl->wl_lds = 1;                               // small tiles: keep the weights beside them
    off += d * pl->wp;
  }
  pl->off_nbr = off;
  pl->off_xs = off;
  int fixed = off * 4;
  const int xs_full = pl->mfma == 2 ? pl->xrows * pl->xp * 4 : (pl->mfma ? n * pl->xp * 4 : n * pl->wp * 4);
  if (fixed > LDS_MAX) return GDN_ERR_UNSUPPORTED;
  // neighbour lists go to LDS when they fit beside the x tile (MFMA path: the whole window)
  int remaining = LDS_MAX - fixed;
  const int xs_min = mode == MODE_ATTN ? 0 : (pl->mfma ? xs_full : (mode == MODE_FUSED ? 16 * pl->wp * 4 : 0));
  if (mode != MODE_PROJECT && nbr_bytes <= remaining - xs_min) {
    pl->nbr_lds = 1;
    pl->off_xs = pl->off_nbr + nbr_bytes / 4;
    remaining -= nbr_bytes;
  }
  if (mode != MODE_ATTN) {
    if (pl->mfma == 2) {
      remaining -= xs_full;                      // (xrows = the chunk, chosen above to fit)
    } else if (xs_full <= remaining) {
      pl->xrows = n;
      remaining -= xs_full;
    } else {
      pl->xrows = (remaining / (pl->wp * 4)) & ~15;
      if (pl->xrows < 16) return GDN_ERR_UNSUPPORTED;
      remaining -= pl->xrows * pl->wp * 4;
    }
  }
  pl->lds_bytes = LDS_MAX - remaining;
  return GDN_OK;
}

template <int D, int MODE, int NT, int PROJ, int LST>
int launch_window(const Plan& pl, const Args& a, hipStream_t stream) {
  constexpr int threads = NT;
  auto kern = gdn_window_kernel<D, MODE, NT, PROJ, LST>;
  const int occ = gdn_blocks_per_cu(reinterpret_cast<const void*>(kern), threads, pl.lds_bytes);
  // every workgroup pays a prologue (neighbour lists, weights, constants): give each at least
  // GDN_MIN_WINDOWS_PER_WG windows when the launch is small (concurrent launches on other streams
  // fill the remaining slots)
  const int min_wpw = max(1, GDN_ENV_INT_ONCE("GDN_MIN_WINDOWS_PER_WG", 1));
  const int grid = max(1, min((pl.batch + min_wpw - 1) / min_wpw, gdn_cu_count() * occ));
  hipLaunchKernelGGL(kern, dim3(grid, pl.nslices), dim3(threads), pl.lds_bytes, stream, pl, a);
  return gdn_launch_status();
}

// runtime -> compile-time selection, one level per parameter
template <int D, int MODE, int NT, int PROJ>
int select_maxr(const Plan& pl, const Args& a, hipStream_t st) {
  if constexpr (MODE == MODE_PROJECT) {
    return launch_window<D, MODE, NT, PROJ, 0>(pl, a, st);
  } else {
    if (pl.nbr_lds) {
      if (pl.pitch == 16) return launch_window<D, MODE, NT, PROJ, 1>(pl, a, st);
      if (pl.pitch == 32) return launch_window<D, MODE, NT, PROJ, 2>(pl, a, st);
      if (pl.pitch <= 80) return launch_window<D, MODE, NT, PROJ, 5>(pl, a, st);
    } else if (pl.pitch <= 80) {
      return launch_window<D, MODE, NT, PROJ, 6>(pl, a, st);
    }
    return launch_window<D, MODE, NT, PROJ, 0>(pl, a, st);
  }
}

template <int D, int MODE, int NT>
int select_proj(const Plan& pl, const Args& a, int threads, hipStream_t st) {
  if constexpr (NT == 1024) {      // only the chunked matrix-core projection runs 16-wave workgroups
    if constexpr (MODE != MODE_ATTN && D >= 32) {
      if (pl.mfma == 2) return select_maxr<D, MODE, NT, 5>(pl, a, st);
    }
    return GDN_ERR_UNSUPPORTED;
  } else if constexpr (MODE == MODE_ATTN) {
    return select_maxr<D, MODE, NT, 1>(pl, a, st);
  } else {
    if constexpr (D >= 32) {
      if (pl.mfma == 2) {      // chunked: x in registers, the LDS x tile one row chunk at a time (512 / 1024 threads)
        if constexpr (NT == 512) return select_maxr<D, MODE, NT, 5>(pl, a, st);
        else return GDN_ERR_UNSUPPORTED;
      }
      if (pl.mfma) {
        if (pl.wpm == 32) return select_maxr<D, MODE, NT, 4>(pl, a, st);
        if (pl.n * pl.w <= 8 * threads) return select_maxr<D, MODE, NT, 2>(pl, a, st);
        return select_maxr<D, MODE, NT, 3>(pl, a, st);
      }
    }
    if (pl.wp == 8) return select_maxr<D, MODE, NT, 0>(pl, a, st);
    return select_maxr<D, MODE, NT, 1>(pl, a, st);
  }
}

template <int MODE>
int dispatch_window(const Plan& pl, const Args& a, int threads, hipStream_t stream) {
#define GDN_CASE(DD)                                                              \
  case DD:                                                                        \
    if (threads == 256) return select_proj<DD, MODE, 256>(pl, a, threads, stream); \
    if (threads == 1024) return select_proj<DD, MODE, 1024>(pl, a, threads, stream); \
    return select_proj<DD, MODE, 512>(pl, a, threads, stream);
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

// 1 when the staged forward (gdn_project_fwd + gdn_attn_aggregate_fwd) takes this shape
int gdn_forward_staged_ok(int n, int w, int d, int k) {
  Plan pl; int threads;
  return make_plan(MODE_PROJECT, 1, n, w, d, 0, &pl, &threads) == GDN_OK &&
         make_plan(MODE_ATTN, 1, n, 0, d, k, &pl, &threads) == GDN_OK;
}

static int project_fwd_impl(const float* x, const float* lin_w, const float* node_terms, int batch,
                            int n, int w, int d, float* xlin, float* s_i, float* s_j, void* stream, bool wide) {
  if (!x || !lin_w || !node_terms || !xlin || !s_i || !s_j) return GDN_ERR_ARG;
  if (!wide && batch > 0 && gdn_use_dense_path() && gdn_dense_supported(n, w, d, 1))
    return gdn_dense_project(x, 0, lin_w, node_terms, batch, n, w, d, xlin, s_i, s_j, (hipStream_t)stream);
  Plan pl; int threads;
  const int rc = make_plan(MODE_PROJECT, batch, n, w, d, 0, &pl, &threads);
  if (rc != GDN_OK) return rc;
  Args a = {};
  a.x = x; a.lin_w = lin_w; a.node_terms = node_terms;
  a.xlin_out = xlin; a.si_out = s_i; a.sj_out = s_j;
  return dispatch_window<MODE_PROJECT>(pl, a, threads, (hipStream_t)stream);
}

extern "C" int gdn_project_fwd(const float* x, const float* lin_w, const float* node_terms, int batch,
                               int n, int w, int d, float* xlin, float* s_i, float* s_j, void* stream) {
  return project_fwd_impl(x, lin_w, node_terms, batch, n, w, d, xlin, s_i, s_j, stream, false);
}
// `_wide`: the fp32 row-gather kernels at every shape — inputs beyond the range of the 16-bit operand terms
extern "C" int gdn_project_fwd_wide(const float* x, const float* lin_w, const float* node_terms, int batch,
                                    int n, int w, int d, float* xlin, float* s_i, float* s_j, void* stream) {
  return project_fwd_impl(x, lin_w, node_terms, batch, n, w, d, xlin, s_i, s_j, stream, true);
}

static int attn_aggregate_fwd_impl(const float* xlin, const float* s_i, const float* s_j,
                                   const uint16_t* nbr, const int32_t* deg, const float* bias,
                                   int batch, int n, int d, int k, float* z, float* alpha,
                                   void* stream, bool wide) {
  if (!xlin || !s_i || !s_j || !nbr || !deg || !bias || !z) return GDN_ERR_ARG;
  if (!wide && batch > 0 && gdn_use_dense_path() && gdn_dense_supported(n, 1, d, k))
    return gdn_dense_attn_aggregate(xlin, 0, s_i, s_j, nbr, bias, batch, n, d, k, z, alpha, (hipStream_t)stream);
  Plan pl; int threads;
  const int rc = make_plan(MODE_ATTN, batch, n, 0, d, k, &pl, &threads);
  if (rc != GDN_OK) return rc;
  Args a = {};
  a.xlin_in = xlin; a.si_in = s_i; a.sj_in = s_j; a.nbr = nbr; a.deg = deg; a.gnn_bias = bias;
  a.z = z; a.alpha = alpha;
  return dispatch_window<MODE_ATTN>(pl, a, threads, (hipStream_t)stream);
}

extern "C" int gdn_attn_aggregate_fwd(const float* xlin, const float* s_i, const float* s_j,
                                      const uint16_t* nbr, const int32_t* deg, const float* bias,
                                      int batch, int n, int d, int k, float* z, float* alpha,
                                      void* stream) {
  return attn_aggregate_fwd_impl(xlin, s_i, s_j, nbr, deg, bias, batch, n, d, k, z, alpha, stream, false);
}
extern "C" int gdn_attn_aggregate_fwd_wide(const float* xlin, const float* s_i, const float* s_j,
                                           const uint16_t* nbr, const int32_t* deg, const float* bias,
                                           int batch, int n, int d, int k, float* z, float* alpha,
                                           void* stream) {
  return attn_aggregate_fwd_impl(xlin, s_i, s_j, nbr, deg, bias, batch, n, d, k, z, alpha, stream, true);
}

namespace {
// sparse-gather kernel for the shapes the dense kernels do not cover (x_bf16: bf16 storage of x and xlin)
int fused_gather(const void* x, int x_bf16, const float* lin_w, const float* node_terms, const uint16_t* nbr,
                 const int32_t* deg, const float* gnn_bias, const float* emb, const float* bn1_affine,
                 const float* bn2_affine, const float* out_w, const float* out_b, int batch, int n, int w,
                 int d, int k, float* out, void* stream, int* gate = nullptr) {
  Plan pl; int threads;
  const int rc = make_plan(MODE_FUSED, batch, n, w, d, k, &pl, &threads);
  if (rc != GDN_OK) return rc;
  if (gate && pl.nslices > 1) return GDN_ERR_UNSUPPORTED;   // (the sliced form clears `out` first: not gateable)
  Args a = {};
  a.gate = gate;
  a.x = static_cast<const float*>(x); a.x_bf16 = x_bf16; a.lin_w = lin_w; a.node_terms = node_terms; a.nbr = nbr; a.deg = deg;
  a.gnn_bias = gnn_bias; a.emb = emb; a.bn1 = bn1_affine; a.bn2 = bn2_affine;
  a.out_w = out_w; a.out_b = out_b; a.out = out;
  if (pl.nslices > 1 &&   // the slices add their partial head outputs into `out`
      hipMemsetAsync(out, 0, (size_t)batch * n * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return GDN_ERR_LAUNCH;
  return dispatch_window<MODE_FUSED>(pl, a, threads, (hipStream_t)stream);
}
}  // namespace

extern "C" int gdn_forward_fused(const float* x, const float* lin_w, const float* node_terms,
                                 const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                                 const float* emb, const float* bn1_affine, const float* bn2_affine,
                                 const float* out_w, const float* out_b, int batch, int n, int w, int d,
                                 int k, float* out, void* stream) {
  if (!x || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine ||
      !bn2_affine || !out_w || !out_b || !out)
    return GDN_ERR_ARG;
  if (batch > 0 && gdn_use_dense_path() && gdn_dense_fused_supported(n, w, d, k))
    return gdn_dense_forward_fused(x, 0, 0, 0, lin_w, node_terms, nbr, gnn_bias, emb, bn1_affine, bn2_affine,
                                   out_w, out_b, batch, n, w, d, k, out, (hipStream_t)stream);
  return fused_gather(x, 0, lin_w, node_terms, nbr, deg, gnn_bias, emb, bn1_affine, bn2_affine, out_w, out_b,
                      batch, n, w, d, k, out, stream);
}

// The row-gather (fp32 VALU) fused forward, optionally GATED on a range guard (include/gdn_hip.h "range guard"):
// guard == null runs it unconditionally (inputs known to exceed the 16-bit operand range); otherwise the launch
// does nothing unless guard[0] != 0, recomputes every window in fp32 when it is, and leaves guard[0..1] = 0.
extern "C" int gdn_forward_fused_gated(int* guard, const float* x, const float* lin_w, const float* node_terms,
                                       const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                                       const float* emb, const float* bn1_affine, const float* bn2_affine,
                                       const float* out_w, const float* out_b, int batch, int n, int w, int d,
                                       int k, float* out, void* stream) {
  if (!x || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine ||
      !bn2_affine || !out_w || !out_b || !out)
    return GDN_ERR_ARG;
  return fused_gather(x, 0, lin_w, node_terms, nbr, deg, gnn_bias, emb, bn1_affine, bn2_affine, out_w, out_b,
                      batch, n, w, d, k, out, stream, guard);
}

static int fused_gather_series(const float* series, int series_len, int first, const float* lin_w,
                               const float* node_terms, const uint16_t* nbr, const int32_t* deg,
                               const float* gnn_bias, const float* emb, const float* bn1_affine,
                               const float* bn2_affine, const float* out_w, const float* out_b,
                               int batch, int n, int w, int d, int k, float* out, void* stream, int* gate);

extern "C" int gdn_forward_fused_series_gated(int* guard, const float* series, int series_len, int first,
                                              const float* lin_w, const float* node_terms, const uint16_t* nbr,
                                              const int32_t* deg, const float* gnn_bias, const float* emb,
                                              const float* bn1_affine, const float* bn2_affine, const float* out_w,
                                              const float* out_b, int batch, int n, int w, int d, int k, float* out,
                                              void* stream) {
  if (!series || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine ||
      !bn2_affine || !out_w || !out_b || !out || series_len <= 0 || first < 0)
    return GDN_ERR_ARG;
  if ((long long)first + batch - 1 + w > series_len) return GDN_ERR_ARG;
  return fused_gather_series(series, series_len, first, lin_w, node_terms, nbr, deg, gnn_bias, emb, bn1_affine,
                             bn2_affine, out_w, out_b, batch, n, w, d, k, out, stream, guard);
}

extern "C" int gdn_forward_fused_series(const float* series, int series_len, int first, const float* lin_w,
                                        const float* node_terms, const uint16_t* nbr, const int32_t* deg,
                                        const float* gnn_bias, const float* emb, const float* bn1_affine,
                                        const float* bn2_affine, const float* out_w, const float* out_b,
                                        int batch, int n, int w, int d, int k, float* out, void* stream) {
  if (!series || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine ||
      !bn2_affine || !out_w || !out_b || !out || series_len <= 0 || first < 0)
    return GDN_ERR_ARG;
  if ((long long)first + batch - 1 + w > series_len) return GDN_ERR_ARG;   // last window must fit
  if (batch > 0 && gdn_use_dense_path() && gdn_dense_fused_supported(n, w, d, k))
    return gdn_dense_forward_fused(series, 0, series_len, first, lin_w, node_terms, nbr, gnn_bias, emb, bn1_affine,
                                   bn2_affine, out_w, out_b, batch, n, w, d, k, out, (hipStream_t)stream);
  return fused_gather_series(series, series_len, first, lin_w, node_terms, nbr, deg, gnn_bias, emb, bn1_affine,
                             bn2_affine, out_w, out_b, batch, n, w, d, k, out, stream, nullptr);
}

static int fused_gather_series(const float* series, int series_len, int first, const float* lin_w,
                               const float* node_terms, const uint16_t* nbr, const int32_t* deg,
                               const float* gnn_bias, const float* emb, const float* bn1_affine,
                               const float* bn2_affine, const float* out_w, const float* out_b,
                               int batch, int n, int w, int d, int k, float* out, void* stream, int* gate) {
  Plan pl; int threads;
  const int rc = make_plan(MODE_FUSED, batch, n, w, d, k, &pl, &threads);
  if (rc != GDN_OK) return rc;
  if (!pl.mfma) return GDN_ERR_UNSUPPORTED;   // series addressing lives in the MFMA-projection variants
  if (gate && pl.nslices > 1) return GDN_ERR_UNSUPPORTED;
  Args a = {};
  a.gate = gate;
  a.x = series; a.series_len = series_len; a.series_first = first;
  a.lin_w = lin_w; a.node_terms = node_terms; a.nbr = nbr; a.deg = deg;
  a.gnn_bias = gnn_bias; a.emb = emb; a.bn1 = bn1_affine; a.bn2 = bn2_affine;
  a.out_w = out_w; a.out_b = out_b; a.out = out;
  if (pl.nslices > 1 &&
      hipMemsetAsync(out, 0, (size_t)batch * n * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return GDN_ERR_LAUNCH;
  return dispatch_window<MODE_FUSED>(pl, a, threads, (hipStream_t)stream);
}

template <typename ZT>
static int head_launch(const ZT* z, const float* emb, const float* bn1_affine, const float* bn2_affine,
                       const float* out_w, const float* out_b, int batch, int n, int d, float* out, float* h2,
                       void* stream) {
  if (!z || !emb || !bn1_affine || !bn2_affine || !out_w || !out_b || !out || batch <= 0 || n <= 0)
    return GDN_ERR_ARG;
  const int rows = batch * n;
  const int grid = min((rows + 63) / 64, gdn_cu_count() * 8);
  hipStream_t st = (hipStream_t)stream;
  switch (d) {
    case 16: hipLaunchKernelGGL((gdn_head_kernel<16, ZT>), dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 32: hipLaunchKernelGGL((gdn_head_kernel<32, ZT>), dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 64: hipLaunchKernelGGL((gdn_head_kernel<64, ZT>), dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    case 128: hipLaunchKernelGGL((gdn_head_kernel<128, ZT>), dim3(grid), dim3(256), 0, st, z, emb, bn1_affine, bn2_affine, out_w, out_b, rows, n, out, h2); break;
    default: return GDN_ERR_UNSUPPORTED;
  }
  return gdn_launch_status();
}

extern "C" int gdn_head_fwd(const float* z, const float* emb, const float* bn1_affine,
                            const float* bn2_affine, const float* out_w, const float* out_b, int batch,
                            int n, int d, float* out, float* h2, void* stream) {
  return head_launch<float>(z, emb, bn1_affine, bn2_affine, out_w, out_b, batch, n, d, out, h2, stream);
}

extern "C" int gdn_head_fwd_bf16(const uint16_t* z, const float* emb, const float* bn1_affine,
                                 const float* bn2_affine, const float* out_w, const float* out_b, int batch,
                                 int n, int d, float* out, float* h2, void* stream) {
  return head_launch<uint16_t>(z, emb, bn1_affine, bn2_affine, out_w, out_b, batch, n, d, out, h2, stream);
}

// ---- bf16 storage (BASELINE configs[2] / [4]): x, xlin and z are bf16 in HBM; logits, softmax and every
// accumulation stay fp32.  Matrix-core path only (n <= 127, d = 64): anything else is refused.
extern "C" int gdn_project_fwd_bf16(const uint16_t* x, const float* lin_w, const float* node_terms, int batch,
                                    int n, int w, int d, uint16_t* xlin, float* s_i, float* s_j, void* stream) {
  if (!x || !lin_w || !node_terms || !xlin || !s_i || !s_j || batch <= 0) return GDN_ERR_ARG;
  return gdn_dense_project(x, 1, lin_w, node_terms, batch, n, w, d, xlin, s_i, s_j, (hipStream_t)stream);
}

extern "C" int gdn_attn_aggregate_fwd_bf16(const uint16_t* xlin, const float* s_i, const float* s_j,
                                           const uint16_t* nbr, const int32_t* deg, const float* bias,
                                           int batch, int n, int d, int k, uint16_t* z, float* alpha,
                                           void* stream) {
  if (!xlin || !s_i || !s_j || !nbr || !deg || !bias || !z || batch <= 0) return GDN_ERR_ARG;
  return gdn_dense_attn_aggregate(xlin, 1, s_i, s_j, nbr, bias, batch, n, d, k, z, alpha, (hipStream_t)stream);
}

extern "C" int gdn_forward_fused_bf16(const uint16_t* x, const float* lin_w, const float* node_terms,
                                      const uint16_t* nbr, const int32_t* deg, const float* gnn_bias,
                                      const float* emb, const float* bn1_affine, const float* bn2_affine,
                                      const float* out_w, const float* out_b, int batch, int n, int w, int d,
                                      int k, float* out, void* stream) {
  if (!x || !lin_w || !node_terms || !nbr || !deg || !gnn_bias || !emb || !bn1_affine || !bn2_affine ||
      !out_w || !out_b || !out || batch <= 0)
    return GDN_ERR_ARG;
  if (gdn_use_dense_path() && gdn_dense_fused_supported(n, w, d, k))
    return gdn_dense_forward_fused(x, 1, 0, 0, lin_w, node_terms, nbr, gnn_bias, emb, bn1_affine, bn2_affine,
                                   out_w, out_b, batch, n, w, d, k, out, (hipStream_t)stream);
  return fused_gather(x, 1, lin_w, node_terms, nbr, deg, gnn_bias, emb, bn1_affine, bn2_affine, out_w, out_b,
                      batch, n, w, d, k, out, stream);
}
