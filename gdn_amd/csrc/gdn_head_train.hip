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
// embedding gradient, d_z).  Column reductions are carried in fp64 per thread and across the workgroup
// (LDS, fixed order).
//
// [r3] The sums ACROSS workgroups are exact: a workgroup's fp64 column partial is split into the limbs of a
// 260-bit fixed-point number (5 x 52 bits, lowest bit 2^-130: everything a product of two fp32 values can be) and
// added limb by limb with 64-bit INTEGER atomics.  Integer addition is associative, so the totals — and with them
// every statistic, gradient and parameter of a training step — do not depend on the order the workgroups finish
// in: bitwise reproducible by construction (fp64 atomics, rounds 1-2, left 1e-16 relative noise in the sums, which
// var = E[z^2] - E[z]^2 amplifies when a channel's mean dwarfs its spread).  A row of totals is converted back to
// fp64 by the first pass that consumes it and kept for the later ones.  Price: two integer atomics per value
// instead of one fp64 atomic and ~1.5 us of prologue in four passes: 0.169 -> 0.181 ms per 512-window step.
//
// Thread layout: a row (one sensor of one window) is covered by LPR = d/4 consecutive lanes holding
// four columns each.  A workgroup takes SLOTS = 256/LPR consecutive sensors and a part of the batch;
// lane group `slot` owns one sensor and walks the windows, four in flight, so the sensor's embedding
// row and its embedding-gradient accumulator stay in registers and need no LDS or atomics per row.
#include "gdn_common.hpp"

namespace {

template <int D>
struct HG {
  static constexpr int LPR = D / 4;
  static constexpr int SLOTS = 256 / LPR;
};

#define GDN_HEAD_REPL 4       // replicas of every column accumulator: same-address atomics serialise
#define GDN_FX_LIMBS 5        // fixed-point accumulator: limbs of 52 bits, limb k = bits 52k .. 52k+51 above 2^-130
#define GDN_FX_WORDS 6        // + one flag word (NaN / inf / beyond 2^130 seen: the total reads as NaN)
#define GDN_FX_LSB 130
#define GDN_HEAD_EMB_PARTS 64  // batch parts of the embedding-gradient pass, one [n,d] partial each

struct RunningStats {   // BatchNorm buffers updated by the forward (any pointer may be null)
  float *rm1, *rv1, *rm2, *rv2;
  long long *nbt1, *nbt2;
  float mom1, mom2;
};

// Counter-based dropout draw (models/GDN.py:114,182 nn.Dropout): element e of step t is kept iff
// mix32(e, seed, t) >= p * 2^32.  Stateless, so the forward pass and the three backward passes of a step
// regenerate the same mask from (seed, step) instead of reading one from HBM; the step counter lives in
// device memory (the optimizer increments it), so a replayed HIP graph draws a fresh mask every replay.
// Not torch's Philox stream: parity tests inject an explicit mask instead.
__device__ __forceinline__ unsigned gdn_mix32(unsigned e, unsigned k0, unsigned k1) {
  unsigned x = e * 0x9E3779B1u + k0;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x += k1; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}

struct HeadArgs {
  const float *z, *emb, *g1, *b1, *g2, *b2, *w, *bo, *mask, *d_out;
  const uint8_t* keep;   // alternative to `mask`: 1 = kept, 0 = dropped, value = keep * keep_scale
  float keep_scale;
  const long long* rng;  // third alternative: {seed, step} in device memory -> the mask is DRAWN here (no tensor)
  unsigned rng_threshold;   // element dropped when its 32-bit hash < threshold (= p * 2^32)
  RunningStats run;
  // an accumulator block of R rows = [R][d] fp64 totals (filled in as passes convert them), then the
  // [REPL][R][FX_WORDS][d] 64-bit accumulator words
  const double* fstats;  // R = 4: sum z, sum z^2, sum h1, sum h1^2
  double* acc;           // forward passes: fstats (writable); backward: R = 6 (head of the workspace)
  float* demb_part;      // backward: [EMB_PARTS][n][d] per-part sums of d_emb
  float *out, *d_z;
  // forward, fused loss (train.py:20-23 F.mse_loss + the first step of loss.backward()): when y is given the
  // last forward pass also writes d_out = 2 (out - y) / count and reduces the loss (per-workgroup fp64 partials,
  // the last workgroup to finish adds them in a fixed order: gdn_mse_kernel's scheme, one launch less)
  const float* y;        // [BN]
  float* mse_d_out;      // [BN]
  double* mse_ws;        // [1 + grid]: ticket, partials
  float* loss;           // [1]
  float* act;            // forward, MLP head (out_layer_num > 1): the [BN, d] activation after dropout instead of `out`
  const float* d_act;    // backward, MLP head: its gradient instead of d_out (x) lin.weight
  int batch, n;
  float eps1, eps2;
  int chunks, parts;     // grid = chunks (of SLOTS sensors) x parts (of the batch)
};

struct BnCols {
  float mu[4], is[4], sc[4], be[4];
};

// ---- exact accumulation across workgroups (see the header) --------------------------------------------------
// An accumulator is GDN_FX_WORDS 64-bit words `stride` words apart (one per limb, then the flag).  Every limb
// receives at most one addend below 2^52 per call: 2048 calls between two resets cannot overflow its 63 bits.
__device__ __forceinline__ void fx_atomic_add(unsigned long long* p, int stride, double x) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
  const int ex = (int)((bits >> 52) & 0x7ffull);
  if (ex == 0) return;                                        // 0 (or below 2^-1022)
  unsigned long long* flag = p + (size_t)GDN_FX_LIMBS * stride;
  if (ex == 0x7ff) { atomicOr(flag, 1ull); return; }          // NaN / inf
  unsigned long long mant = (bits & ((1ull << 52) - 1ull)) | (1ull << 52);
  int pos = ex - 1075 + GDN_FX_LSB;                           // position of the mantissa's lowest bit
  if (pos < 0) {                                              // bits below 2^-130 are dropped (towards zero)
    if (pos <= -53) return;
    mant >>= -pos;
    pos = 0;
  }
  const int k = pos / 52, o = pos - 52 * k;
  const unsigned __int128 v = (unsigned __int128)mant << o;   // < 2^104: two limbs
  long long l0 = (long long)(unsigned long long)(v & (((unsigned __int128)1 << 52) - 1));
  long long l1 = (long long)(unsigned long long)(v >> 52);
  if (k >= GDN_FX_LIMBS || (l1 != 0 && k + 1 >= GDN_FX_LIMBS)) { atomicOr(flag, 1ull); return; }
  if (bits >> 63) { l0 = -l0; l1 = -l1; }
  if (l0 != 0) atomicAdd(p + (size_t)k * stride, (unsigned long long)l0);
  if (l1 != 0) atomicAdd(p + (size_t)(k + 1) * stride, (unsigned long long)l1);
}

// value of an accumulator summed over its replicas (`repl_stride` words apart): limb sums are exact integers,
// carries are propagated, the magnitude is converted limb by limb from the top (no cancellation)
__device__ __forceinline__ double fx_total(const unsigned long long* p, int stride, size_t repl_stride) {
  long long l[GDN_FX_LIMBS];
  unsigned long long flag = 0ull;
  unsigned long long w[GDN_HEAD_REPL][GDN_FX_WORDS];
#pragma unroll
  for (int r = 0; r < GDN_HEAD_REPL; ++r)
#pragma unroll
    for (int k = 0; k < GDN_FX_WORDS; ++k) w[r][k] = p[r * repl_stride + (size_t)k * stride];   // all loads in flight
#pragma unroll
  for (int k = 0; k < GDN_FX_LIMBS; ++k) l[k] = 0;
#pragma unroll
  for (int r = 0; r < GDN_HEAD_REPL; ++r) {
#pragma unroll
    for (int k = 0; k < GDN_FX_LIMBS; ++k) l[k] += (long long)w[r][k];
    flag |= w[r][GDN_FX_LIMBS];
  }
  if (flag) return __longlong_as_double(0x7ff8000000000000ll);
  auto carry = [&]() {
#pragma unroll
    for (int k = 0; k + 1 < GDN_FX_LIMBS; ++k) {
      const long long c = l[k] >> 52;            // floor: the limb lands in [0, 2^52)
      l[k] -= c << 52;
      l[k + 1] += c;
    }
  };
  carry();
  const bool neg = l[GDN_FX_LIMBS - 1] < 0;
  if (neg) {
#pragma unroll
    for (int k = 0; k < GDN_FX_LIMBS; ++k) l[k] = -l[k];
    carry();
  }
  double s = 0.0;
#pragma unroll
  for (int k = GDN_FX_LIMBS - 1; k >= 0; --k)     // exact scaling: the factor is the double 2^(52k - 130)
    s += (double)l[k] * __longlong_as_double((long long)(1023 + 52 * k - GDN_FX_LSB) << 52);
  return neg ? -s : s;
}
// word offset of limb 0 of accumulator (replica, row, column) in a block of `rows` rows
__host__ __device__ __forceinline__ size_t fx_at(int repl, int rows, int row, int d, int col) {
  return (size_t)rows * d + (((size_t)repl * rows + row) * GDN_FX_WORDS) * d + col;
}
__host__ __device__ __forceinline__ size_t fx_block_words(int rows, int d) {
  return (size_t)rows * d + (size_t)GDN_HEAD_REPL * rows * GDN_FX_WORDS * d;
}

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// sum the per-thread column partials over the workgroup's SLOTS lane groups, then one fp64 atomic per
// column per workgroup
template <int D>
__device__ __forceinline__ void col_reduce(const double (&v)[4], double* red, unsigned long long* gdst, int tid,
                                           int slot, int c0) {
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) red[slot * D + c0 + q] = v[q];
  __syncthreads();
  if (tid < D) {
    double s = 0.0;
    for (int q = 0; q < HG<D>::SLOTS; ++q) s += red[q * D + tid];
    fx_atomic_add(gdst + tid, D, s);
  }
}

enum { H_STAT1 = 0, H_STAT2 = 1, H_OUT = 2, H_BWD2 = 3, H_BWD1 = 4, H_DZ = 5 };

#define GDN_HEAD_UNROLL 4   // windows in flight per thread: the passes are latency bound otherwise

// Work split: workgroup = (sensor chunk, batch part).  Lane group `slot` owns ONE sensor
// n = chunk*SLOTS + slot and walks the windows of the part, so the embedding row and the
// embedding-gradient accumulator of (n, columns) live in registers.
template <int D, int MODE>
__global__ __launch_bounds__(256) void gdn_head_train_kernel(const HeadArgs a) {
  using G = HG<D>;
  constexpr int U = GDN_HEAD_UNROLL;
  __shared__ double red[1024];                                // [SLOTS][D]
  const int tid = threadIdx.x, lr = tid % G::LPR, slot = tid / G::LPR, c0 = lr * 4;
  const double rows = (double)a.batch * (double)a.n;
  const int chunk = blockIdx.x % a.chunks, part = blockIdx.x / a.chunks;
  const int n = chunk * G::SLOTS + slot;
  const bool live = n < a.n;
  const int b0 = (int)((long long)a.batch * part / a.parts);
  const int b1 = (int)((long long)a.batch * (part + 1) / a.parts);

  // (Measured, round 2: issuing the first round of row loads BEFORE the statistics prologue below, or 8 rows
  // in flight instead of 4, made the backward passes 1-6 us slower — more live registers — and the forward
  // ones ~1 us faster: a wash, not kept.)
  unsigned rng_k0 = 0, rng_k1 = 0;
  if (MODE >= H_OUT && a.rng) {
    const unsigned long long seed = (unsigned long long)a.rng[0], step = (unsigned long long)a.rng[1];
    rng_k0 = (unsigned)seed ^ (unsigned)(step * 0x9E3779B97F4A7C15ull >> 32);
    rng_k1 = (unsigned)(seed >> 32) + (unsigned)step * 0x7F4A7C15u;
  }
  float e[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE >= H_STAT2 && live) ld4(a.emb + (size_t)n * D + c0, e);
  float zq[U][4], mq[U][4], goq[U], gaq[U][4];
  auto load_round = [&](int bq) {
#pragma unroll
    for (int u = 0; u < U; ++u) {                     // all loads of the round first
      const int b = min(bq + u, b1 - 1);
      const size_t row = (size_t)b * a.n + n;
      ld4(a.z + row * D + c0, zq[u]);
      mq[u][0] = mq[u][1] = mq[u][2] = mq[u][3] = 1.f;
      if (MODE >= H_OUT && a.mask) {
        ld4(a.mask + row * D + c0, mq[u]);
      } else if (MODE >= H_OUT && a.keep) {           // one byte per element: a quarter of the mask traffic
        const uchar4 kb = *reinterpret_cast<const uchar4*>(a.keep + row * D + c0);
        mq[u][0] = kb.x * a.keep_scale; mq[u][1] = kb.y * a.keep_scale;
        mq[u][2] = kb.z * a.keep_scale; mq[u][3] = kb.w * a.keep_scale;
      } else if (MODE >= H_OUT && a.rng) {            // drawn in place: no mask traffic at all
        const unsigned e0 = (unsigned)(row * D + c0);
#pragma unroll
        for (int v = 0; v < 4; ++v) mq[u][v] = gdn_mix32(e0 + v, rng_k0, rng_k1) >= a.rng_threshold ? a.keep_scale : 0.f;
      }
      goq[u] = (MODE >= H_BWD2 && a.d_out) ? a.d_out[row] : 0.f;
      gaq[u][0] = gaq[u][1] = gaq[u][2] = gaq[u][3] = 0.f;
      if (MODE >= H_BWD2 && a.d_act) ld4(a.d_act + row * D + c0, gaq[u]);
    }
  };

  // totals of the accumulators earlier passes left behind (summed over their replicas once per workgroup):
  // rows 0-3 = column sums of z, z^2, h1, h1^2; rows 4-7 = sums of d_y2, d_y2*xhat2, d_y1, d_y1*xhat1
  __shared__ double tot[8 * D];
  constexpr int NF = MODE >= H_OUT ? 4 : (MODE >= H_STAT2 ? 2 : 0);
  constexpr int NB = MODE == H_DZ ? 4 : (MODE == H_BWD1 ? 2 : 0);
  const unsigned long long* fwords = reinterpret_cast<const unsigned long long*>(a.fstats);
  unsigned long long* awords = reinterpret_cast<unsigned long long*>(a.acc);
  // A row of totals is converted from its limbs by the FIRST pass that consumes it (every workgroup of that pass
  // does the same exact integer arithmetic); workgroup 0 leaves the fp64 value in the block's totals, where the
  // later passes read it: 8 row conversions per step instead of 24 (each costs a pass ~1.5 us of prologue).
  //   forward rows 0,1: fresh in H_STAT2; rows 2,3: fresh in H_OUT.  backward rows 0,1: fresh in H_BWD1; rows
  //   2,3: fresh in H_DZ; rows 4,5 (d_lin_w, d_lin_b) are converted by the finish kernel.
  for (int t = tid; t < NF * D; t += 256) {
    const int row = t / D;
    const bool fresh = (MODE == H_STAT2) || (MODE == H_OUT && row >= 2);
    double v;
    if (fresh) {
      v = fx_total(fwords + fx_at(0, 4, row, D, t % D), D, (size_t)4 * GDN_FX_WORDS * D);
      if (blockIdx.x == 0) const_cast<double*>(a.fstats)[t] = v;
    } else {
      v = a.fstats[t];
    }
    tot[t] = v;
  }
  for (int t = tid; t < NB * D; t += 256) {
    const int row = t / D;
    const bool fresh = (MODE == H_BWD1) || (MODE == H_DZ && row >= 2);
    double v;
    if (fresh) {
      v = fx_total(awords + fx_at(0, 6, row, D, t % D), D, (size_t)6 * GDN_FX_WORDS * D);
      if (blockIdx.x == 0) a.acc[t] = v;
    } else {
      v = a.acc[t];
    }
    tot[4 * D + t] = v;
  }
  if constexpr (NF > 0) __syncthreads();

  if constexpr (MODE == H_OUT) {
    // running_mean / running_var / num_batches_tracked of both BatchNorms (torch: momentum update with
    // the UNBIASED batch variance); done once, by workgroup 0 of the last forward pass
    if (blockIdx.x == 0) {
      if (tid < 2 * D) {
        const int which = tid / D, t = tid % D;
        float* rm = which ? a.run.rm2 : a.run.rm1;
        float* rv = which ? a.run.rv2 : a.run.rv1;
        const float mom = which ? a.run.mom2 : a.run.mom1;
        if (rm && rv) {
          const double m = tot[which * 2 * D + t] / rows;
          double var = tot[which * 2 * D + D + t] / rows - m * m;
          if (var < 0.0) var = 0.0;
          rm[t] = (1.f - mom) * rm[t] + mom * (float)m;
          rv[t] = (1.f - mom) * rv[t] + mom * (float)(var * rows / (rows - 1.0));
        }
      }
      if (tid == 0) {
        if (a.run.nbt1) *a.run.nbt1 += 1;
        if (a.run.nbt2) *a.run.nbt2 += 1;
      }
    }
  }

  // Per-column constants ONCE per workgroup (one thread per column and BatchNorm: the fp64 divisions and the
  // square root), handed to the SLOTS lane groups through LDS — computed per thread they were the same ~20 fp64
  // transcendental sequences 16 times over, a couple of microseconds of every latency-bound pass.
  __shared__ float colc[12 * D];      // [mu1 | is1 | sc1 | be1 | mu2 | is2 | sc2 | be2 | m2a | m2b | m1a | m1b]
  if constexpr (MODE >= H_STAT2) {
    constexpr int NBN = MODE >= H_OUT ? 2 : 1;
    for (int t = tid; t < NBN * D; t += 256) {
      const int which = t / D, c = t - which * D;
      const double* sum = tot + which * 2 * D;
      const double m = sum[c] / rows;
      double var = sum[D + c] / rows - m * m;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)(which ? a.eps2 : a.eps1)));
      float* dst = colc + which * 4 * D;
      dst[c] = (float)m;
      dst[D + c] = is;
      dst[2 * D + c] = (which ? a.g2 : a.g1)[c] * is;
      dst[3 * D + c] = (which ? a.b2 : a.b1)[c];
    }
    if constexpr (MODE >= H_BWD1) {
      constexpr int NM = MODE == H_DZ ? 4 : 2;
      for (int t = tid; t < NM * D; t += 256) colc[8 * D + t] = (float)(tot[4 * D + t] / rows);   // means of d_y2, d_y2 xhat2, d_y1, d_y1 xhat1
    }
    __syncthreads();
  }
  BnCols bn1 = {}, bn2 = {};
  if constexpr (MODE >= H_STAT2) {
    ld4(colc + c0, bn1.mu); ld4(colc + D + c0, bn1.is); ld4(colc + 2 * D + c0, bn1.sc); ld4(colc + 3 * D + c0, bn1.be);
  }
  if constexpr (MODE >= H_OUT) {
    ld4(colc + 4 * D + c0, bn2.mu); ld4(colc + 5 * D + c0, bn2.is); ld4(colc + 6 * D + c0, bn2.sc); ld4(colc + 7 * D + c0, bn2.be);
  }
  float w4[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE >= H_OUT && a.w) ld4(a.w + c0, w4);
  float m2a[4] = {}, m2b[4] = {}, m1a[4] = {}, m1b[4] = {};
  if constexpr (MODE >= H_BWD1) {
    ld4(colc + 8 * D + c0, m2a);      // mean of d_y2
    ld4(colc + 9 * D + c0, m2b);      // mean of d_y2 * xhat2
  }
  if constexpr (MODE == H_DZ) {
    ld4(colc + 10 * D + c0, m1a);
    ld4(colc + 11 * D + c0, m1b);
  }
  double acc0[4] = {0.0, 0.0, 0.0, 0.0}, acc1[4] = {0.0, 0.0, 0.0, 0.0}, acc2[4] = {0.0, 0.0, 0.0, 0.0};
  double acc_s = 0.0;
  const float bias_o = (MODE == H_OUT && a.bo) ? a.bo[0] : 0.f;
  const float mse_scale = (float)(2.0 / rows);
  if (live) {
    for (int bq = b0; bq < b1; bq += U) {
      load_round(bq);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (bq + u >= b1) break;
        const size_t row = (size_t)(bq + u) * a.n + n;
        const float(&z)[4] = zq[u];
        const float(&m)[4] = mq[u];
        if constexpr (MODE == H_STAT1) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const double zd = (double)z[v];
            acc0[v] += zd;
            acc1[v] = fma(zd, zd, acc1[v]);
          }
          continue;
        }
        float y1[4], a1[4], h1[4];
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
        float y2[4], a2[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          y2[v] = fmaf(h1[v] - bn2.mu[v], bn2.sc[v], bn2.be[v]);
          a2[v] = fmaxf(y2[v], 0.f);
        }
        if constexpr (MODE == H_OUT) {
          if (a.act) {   // MLP head: hand the dropped-out activation to the OutLayer (models/GDN.py:182-183)
            *reinterpret_cast<float4*>(a.act + row * D + c0) =
                make_float4(a2[0] * m[0], a2[1] * m[1], a2[2] * m[2], a2[3] * m[3]);
            continue;
          }
          float part_o = 0.f;
#pragma unroll
          for (int v = 0; v < 4; ++v) part_o = fmaf(a2[v] * m[v], w4[v], part_o);
#pragma unroll
          for (int s = 1; s < G::LPR; s <<= 1) part_o += __shfl_xor(part_o, s);   // lanes of one row
          if (lr == 0) {
            const float o = part_o + bias_o;
            a.out[row] = o;
            if (a.y) {
              const float df = o - a.y[row];
              acc_s = fma((double)df, (double)df, acc_s);
              a.mse_d_out[row] = df * mse_scale;
            }
          }
          continue;
        }
        const float go = goq[u];
        float dy2[4], x2h[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float gh = a.d_act ? gaq[u][v] : go * w4[v];   // gradient of the head's [BN, d] activation
          dy2[v] = y2[v] > 0.f ? gh * m[v] : 0.f;
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
            acc2[v] += (double)(dh1[v] * a1[v]);        // d_emb[n, column]: private to this thread
          }
          continue;
        }
        if constexpr (MODE == H_DZ) {
          float4 o;
          o.x = bn1.sc[0] * (dy1[0] - m1a[0] - x1h[0] * m1b[0]);
          o.y = bn1.sc[1] * (dy1[1] - m1a[1] - x1h[1] * m1b[1]);
          o.z = bn1.sc[2] * (dy1[2] - m1a[2] - x1h[2] * m1b[2]);
          o.w = bn1.sc[3] * (dy1[3] - m1a[3] - x1h[3] * m1b[3]);
          *reinterpret_cast<float4*>(a.d_z + row * D + c0) = o;
        }
      }
    }
  }

  if constexpr (MODE == H_OUT) {
    if (a.y) {                                    // loss = mean((out - y)^2): fixed-order reduction, last workgroup finishes
      __shared__ bool last;
      __syncthreads();
      red[tid] = acc_s;
      __syncthreads();
      for (int q = 128; q > 0; q >>= 1) {
        if (tid < q) red[tid] += red[tid + q];
        __syncthreads();
      }
      unsigned long long* ticket = reinterpret_cast<unsigned long long*>(a.mse_ws);
      if (tid == 0) {
        // no release/acquire fence (each would write back this XCD's whole L2: ~20 us over 512 workgroups): the
        // partial is an agent-scope atomic store, acknowledged (vmcnt) before the ticket is taken, and the last
        // workgroup reads the partials back with agent-scope atomic loads
        __hip_atomic_store(a.mse_ws + 1 + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == (unsigned long long)gridDim.x - 1ull;
      }
      __syncthreads();
      if (last) {
        double sum = 0.0;
        for (int q = tid; q < (int)gridDim.x; q += 256)
          sum += __hip_atomic_load(a.mse_ws + 1 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[tid] = sum;
        __syncthreads();
        for (int q = 128; q > 0; q >>= 1) {
          if (tid < q) red[tid] += red[tid + q];
          __syncthreads();
        }
        if (tid == 0) {
          a.loss[0] = (float)(red[0] / rows);
          __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  const int repl = blockIdx.x % GDN_HEAD_REPL;
  if constexpr (MODE == H_STAT1 || MODE == H_STAT2) {
    const int r0 = MODE == H_STAT1 ? 0 : 2;
    col_reduce<D>(acc0, red, awords + fx_at(repl, 4, r0, D, 0), tid, slot, c0);
    col_reduce<D>(acc1, red, awords + fx_at(repl, 4, r0 + 1, D, 0), tid, slot, c0);
  }
  if constexpr (MODE == H_BWD2) {
    col_reduce<D>(acc0, red, awords + fx_at(repl, 6, 0, D, 0), tid, slot, c0);
    col_reduce<D>(acc1, red, awords + fx_at(repl, 6, 1, D, 0), tid, slot, c0);
    col_reduce<D>(acc2, red, awords + fx_at(repl, 6, 4, D, 0), tid, slot, c0);
    __syncthreads();
    red[tid] = acc_s;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int q = 0; q < 256; ++q) s += red[q];
      fx_atomic_add(awords + fx_at(repl, 6, 5, D, 0), D, s);
    }
  }
  if constexpr (MODE == H_BWD1) {
    col_reduce<D>(acc0, red, awords + fx_at(repl, 6, 2, D, 0), tid, slot, c0);
    col_reduce<D>(acc1, red, awords + fx_at(repl, 6, 3, D, 0), tid, slot, c0);
    if (live)   // every (part, sensor, column) has exactly one owner: plain store, summed by the finish kernel
      *reinterpret_cast<float4*>(a.demb_part + ((size_t)part * a.n + n) * D + c0) =
          make_float4((float)acc2[0], (float)acc2[1], (float)acc2[2], (float)acc2[3]);
  }
}

__device__ __forceinline__ void head_finish_body(double* __restrict__ ws, const float* __restrict__ demb_part,
                                                 int parts, int n, int d, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                                                 float* d_bn2_b, float* d_lin_w, float* d_lin_b, float* d_emb,
                                                 double* zero_stats, int block) {
  const int t = block * (int)blockDim.x + (int)threadIdx.x;
  if (block == 0) {
    // the six gradient rows of the head, then — every pass of this step is complete, this is the last reader —
    // both accumulator blocks are left zeroed for the next step, which then needs no memset launches
    __shared__ float fin[6 * 128];
    unsigned long long* wsw = reinterpret_cast<unsigned long long*>(ws);
    for (int i = (int)threadIdx.x; i < 6 * d; i += (int)blockDim.x)     // rows 0-3: converted by H_BWD1 / H_DZ
      fin[i] = i < 4 * d ? (float)ws[i]
                         : (float)fx_total(wsw + fx_at(0, 6, i / d, d, i % d), d, (size_t)6 * GDN_FX_WORDS * d);
    __syncthreads();
    if (zero_stats) {
      unsigned long long* zs = reinterpret_cast<unsigned long long*>(zero_stats);
      for (int i = (int)threadIdx.x; i < (int)fx_block_words(4, d); i += (int)blockDim.x) zs[i] = 0ull;
      for (int i = (int)threadIdx.x; i < (int)fx_block_words(6, d); i += (int)blockDim.x) wsw[i] = 0ull;
    }
    for (int c = (int)threadIdx.x; c < d; c += (int)blockDim.x) {
      d_bn2_b[c] = fin[c];
      d_bn2_w[c] = fin[d + c];
      d_bn1_b[c] = fin[2 * d + c];
      d_bn1_w[c] = fin[3 * d + c];
      if (d_lin_w) d_lin_w[c] = fin[4 * d + c];
      if (c == 0 && d_lin_b) d_lin_b[0] = fin[5 * d];
    }
  }
  if (t < n * d) {
    double s = 0.0;
#pragma unroll 8
    for (int p = 0; p < parts; ++p) s += (double)demb_part[(size_t)p * n * d + t];
    d_emb[t] = (float)s;
  }
}

__global__ void gdn_head_finish_kernel(double* __restrict__ ws, const float* __restrict__ demb_part,
                                       int parts, int n, int d, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                                       float* d_bn2_b, float* d_lin_w, float* d_lin_b, float* d_emb,
                                       double* zero_stats) {
  head_finish_body(ws, demb_part, parts, n, d, d_bn1_w, d_bn1_b, d_bn2_w, d_bn2_b, d_lin_w, d_lin_b, d_emb, zero_stats,
                   (int)blockIdx.x);
}

// The training step's two small reductions in ONE launch (a launch costs ~5 us inside a replayed step whatever
// it does): workgroups [0, g1) finish the head backward (gdn_head_finish_kernel), the rest sum the partial rows
// of gdn_project_bwd (gdn_project_reduce_kernel).  Independent work, disjoint outputs.
struct TailArgs {
  double* head_ws; const float* demb_part; int parts, n, d;
  float *d_bn1_w, *d_bn1_b, *d_bn2_w, *d_bn2_b, *d_lin_w, *d_lin_b, *d_emb;
  double* zero_stats;
  int g1;
  const float* proj_part; int rows, w, wp;
  float *d_proj_w, *d_a, *d_c;
};
__global__ __launch_bounds__(256) void gdn_train_tail_kernel(const TailArgs t) {
  if ((int)blockIdx.x < t.g1)
    head_finish_body(t.head_ws, t.demb_part, t.parts, t.n, t.d, t.d_bn1_w, t.d_bn1_b, t.d_bn2_w, t.d_bn2_b, t.d_lin_w,
                     t.d_lin_b, t.d_emb, t.zero_stats, (int)blockIdx.x);
  else
    gdn_project_reduce_body(t.proj_part, t.rows, t.d, t.n, t.w, t.wp, t.d_proj_w, t.d_a, t.d_c, (int)blockIdx.x - t.g1);
}

int head_parts(int batch, int chunks, int mode) {
  // reduction passes: ~2 workgroups per CU (fewer atomics); streaming passes: ~4
  const int target = (mode == H_OUT || mode == H_DZ ? 4 : 2) * gdn_cu_count();
  int parts = (target + chunks - 1) / chunks;
  const int by_rounds = (batch + GDN_HEAD_UNROLL - 1) / GDN_HEAD_UNROLL;   // at least one full round each
  if (parts > by_rounds) parts = by_rounds;
  if (mode == H_BWD1 && parts > GDN_HEAD_EMB_PARTS) parts = GDN_HEAD_EMB_PARTS;
  return parts < 1 ? 1 : parts;
}

template <int D, int MODE>
void launch_pass(HeadArgs a, hipStream_t st) {
  a.chunks = (a.n + HG<D>::SLOTS - 1) / HG<D>::SLOTS;
  a.parts = head_parts(a.batch, a.chunks, MODE);
  hipLaunchKernelGGL((gdn_head_train_kernel<D, MODE>), dim3(a.chunks * a.parts), dim3(256), 0, st, a);
}

// loss = mean((out - y)^2) and its gradient d_out = 2 (out - y) / count in one launch (train.py:20-23
// `F.mse_loss(..., reduction='mean')` + the first step of loss.backward()).  Per-workgroup fp64 partial
// sums; the last workgroup to finish (ticket with release/acquire ordering) adds them up in a fixed
// order, so the loss is bitwise reproducible.  ws[0] = ticket (left at 0), ws[1..grid] = partials.
__global__ __launch_bounds__(256) void gdn_mse_kernel(const float* __restrict__ out, const float* __restrict__ y,
                                                      long long count, float* __restrict__ d_out,
                                                      double* __restrict__ ws, float* __restrict__ loss) {
  __shared__ double red[256];
  __shared__ bool last;
  const int tid = threadIdx.x;
  const float scale = (float)(2.0 / (double)count);
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < count; i += (long long)gridDim.x * 256) {
    const float df = out[i] - y[i];
    acc = fma((double)df, (double)df, acc);
    d_out[i] = df * scale;
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  unsigned long long* ticket = reinterpret_cast<unsigned long long*>(ws);
  if (tid == 0) {
    __hip_atomic_store(ws + 1 + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (unsigned long long)gridDim.x - 1ull;
  }
  __syncthreads();
  if (!last) return;
  double s = 0.0;
  for (int b = tid; b < (int)gridDim.x; b += 256)
    s += __hip_atomic_load(ws + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  red[tid] = s;
  __syncthreads();
  for (int q = 128; q > 0; q >>= 1) {
    if (tid < q) red[tid] += red[tid + q];
    __syncthreads();
  }
  if (tid == 0) {
    loss[0] = (float)(red[0] / (double)count);
    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// gdn_exact_sum: the accumulator above on its own (tests pin it against an exactly rounded CPU sum)
__global__ void gdn_exact_sum_kernel(const double* __restrict__ x, int count, unsigned long long* __restrict__ ws) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) fx_atomic_add(ws + (size_t)(i % GDN_HEAD_REPL) * GDN_FX_WORDS, 1, x[i]);
}
__global__ void gdn_exact_sum_finish_kernel(unsigned long long* __restrict__ ws, double* __restrict__ out) {
  out[0] = fx_total(ws, 1, GDN_FX_WORDS);
  for (int i = 0; i < GDN_HEAD_REPL * GDN_FX_WORDS; ++i) ws[i] = 0ull;
}

#define GDN_MSE_MAX_GRID 256

bool head_shape_ok(int batch, int n, int d) {
  return batch > 0 && n > 0 && n <= 4096 && (long long)batch * n >= 2;
}

}  // namespace

extern "C" long long gdn_exact_sum_workspace_bytes(void) { return (long long)GDN_HEAD_REPL * GDN_FX_WORDS * 8; }

extern "C" int gdn_exact_sum(const double* x, int count, void* workspace, double* out, void* stream) {
  if (!x || !workspace || !out || count <= 0) return GDN_ERR_ARG;
  if (count > 2048) return GDN_ERR_UNSUPPORTED;     // addends per limb between two resets
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gdn_exact_sum_kernel, dim3((count + 63) / 64), dim3(64), 0, st, x, count,
                     static_cast<unsigned long long*>(workspace));
  hipLaunchKernelGGL(gdn_exact_sum_finish_kernel, dim3(1), dim3(1), 0, st, static_cast<unsigned long long*>(workspace), out);
  return gdn_launch_status();
}

extern "C" long long gdn_mse_workspace_bytes(void) { return (GDN_MSE_MAX_GRID + 1) * (long long)sizeof(double); }

extern "C" int gdn_mse_loss_grad(const float* out, const float* y, long long count, double* workspace,
                                 float* loss, float* d_out, void* stream) {
  if (!out || !y || !workspace || !loss || !d_out || count <= 0) return GDN_ERR_ARG;
  long long grid = (count + 1023) / 1024;
  if (grid > GDN_MSE_MAX_GRID) grid = GDN_MSE_MAX_GRID;
  hipLaunchKernelGGL(gdn_mse_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, out, y, count, d_out,
                     workspace, loss);
  return gdn_launch_status();
}

extern "C" long long gdn_head_train_stats_bytes(int d) {
  return d <= 0 ? 0 : (long long)fx_block_words(4, d) * (long long)sizeof(double);
}

extern "C" long long gdn_head_train_workspace_bytes(int n, int d) {
  if (n <= 0 || d <= 0) return 0;
  return (long long)fx_block_words(6, d) * (long long)sizeof(double) +
         (long long)GDN_HEAD_EMB_PARTS * n * d * (long long)sizeof(float);
}

static unsigned drop_threshold(float p_drop) {
  const double t = (double)p_drop * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

static int head_train_fwd_impl(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                  const float* bn2_w, const float* bn2_b, const float* lin_w,
                                  const float* lin_b, const float* mask, const uint8_t* keep, float keep_scale,
                                  const long long* rng, float p_drop, int batch, int n,
                                  int d, float eps1,
                                  float eps2, float momentum1, float momentum2, float* running_mean1,
                                  float* running_var1, long long* batches1, float* running_mean2,
                                  float* running_var2, long long* batches2, double* stats, float* out,
                                  void* stream, float* act = nullptr, bool zeroed = false,
                                  const float* y = nullptr, float* d_out = nullptr, double* mse_ws = nullptr,
                                  float* loss = nullptr) {
  if (!z || !emb || !bn1_w || !bn1_b || !bn2_w || !bn2_b || !stats) return GDN_ERR_ARG;
  if (act ? (lin_w || lin_b || out) : (!lin_w || !lin_b || !out)) return GDN_ERR_ARG;
  if (!head_shape_ok(batch, n, d)) return GDN_ERR_ARG;   // torch: "Expected more than 1 value per channel"
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  HeadArgs a = {};
  a.z = z; a.emb = emb; a.g1 = bn1_w; a.b1 = bn1_b; a.g2 = bn2_w; a.b2 = bn2_b; a.w = lin_w; a.bo = lin_b;
  a.mask = mask; a.keep = mask ? nullptr : keep; a.keep_scale = keep_scale;
  if (!mask && !keep && rng && p_drop > 0.f) {
    a.rng = rng; a.rng_threshold = drop_threshold(p_drop); a.keep_scale = 1.f / (1.f - p_drop);
  }
  a.fstats = stats; a.acc = stats; a.out = out; a.act = act; a.batch = batch; a.n = n;
  a.y = y; a.mse_d_out = d_out; a.mse_ws = mse_ws; a.loss = loss;
  a.eps1 = eps1; a.eps2 = eps2;
  a.run = {running_mean1, running_var1, running_mean2, running_var2, batches1, batches2, momentum1, momentum2};
  if (!zeroed && hipMemsetAsync(stats, 0, fx_block_words(4, d) * sizeof(double), st) != hipSuccess)
    return GDN_ERR_LAUNCH;
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
  return gdn_launch_status();
}

extern "C" int gdn_head_train_fwd(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                  const float* bn2_w, const float* bn2_b, const float* lin_w,
                                  const float* lin_b, const float* mask, const uint8_t* keep, float keep_scale, int batch, int n,
                                  int d, float eps1,
                                  float eps2, float momentum1, float momentum2, float* running_mean1,
                                  float* running_var1, long long* batches1, float* running_mean2,
                                  float* running_var2, long long* batches2, double* stats, float* out,
                                  void* stream) {
  return head_train_fwd_impl(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, lin_b, mask, keep, keep_scale, nullptr, 0.f,
                             batch, n, d, eps1, eps2, momentum1, momentum2, running_mean1, running_var1, batches1,
                             running_mean2, running_var2, batches2, stats, out, stream);
}

extern "C" int gdn_head_train_fwd_rng(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                      const float* bn2_w, const float* bn2_b, const float* lin_w,
                                      const float* lin_b, const long long* rng_seed_step, float p_drop, int batch,
                                      int n, int d, float eps1, float eps2, float momentum1, float momentum2,
                                      float* running_mean1, float* running_var1, long long* batches1,
                                      float* running_mean2, float* running_var2, long long* batches2,
                                      double* stats, float* out, int buffers_zeroed, void* stream) {
  if (!rng_seed_step || p_drop < 0.f || p_drop >= 1.f) return GDN_ERR_ARG;
  return head_train_fwd_impl(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, lin_b, nullptr, nullptr, 1.f, rng_seed_step,
                             p_drop, batch, n, d, eps1, eps2, momentum1, momentum2, running_mean1, running_var1,
                             batches1, running_mean2, running_var2, batches2, stats, out, stream, nullptr,
                             buffers_zeroed != 0);
}

static int head_train_bwd_impl(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                                  const float* bn1_b, const float* bn2_w, const float* bn2_b,
                                  const float* lin_w, const float* mask, const uint8_t* keep, float keep_scale,
                                  const long long* rng, float p_drop,
                                  const double* stats, int batch,
                                  int n, int d, float eps1, float eps2, double* workspace, float* d_z,
                                  float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                                  float* d_bn2_b, float* d_lin_w, float* d_lin_b, void* stream,
                                  const float* d_act = nullptr, bool zeroed = false, bool defer_finish = false) {
  if (!z || !emb || !bn1_w || !bn1_b || !bn2_w || !bn2_b || !stats || !workspace ||
      !d_z || !d_emb || !d_bn1_w || !d_bn1_b || !d_bn2_w || !d_bn2_b)
    return GDN_ERR_ARG;
  if (d_act ? (d_out || lin_w || d_lin_w || d_lin_b) : (!d_out || !lin_w || !d_lin_w || !d_lin_b)) return GDN_ERR_ARG;
  if (!head_shape_ok(batch, n, d)) return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  HeadArgs a = {};
  a.z = z; a.emb = emb; a.g1 = bn1_w; a.b1 = bn1_b; a.g2 = bn2_w; a.b2 = bn2_b; a.w = lin_w; a.bo = nullptr;
  a.mask = mask; a.keep = mask ? nullptr : keep; a.keep_scale = keep_scale;
  if (!mask && !keep && rng && p_drop > 0.f) {
    a.rng = rng; a.rng_threshold = drop_threshold(p_drop); a.keep_scale = 1.f / (1.f - p_drop);
  }
  a.d_out = d_out; a.d_act = d_act; a.fstats = stats; a.acc = workspace; a.d_z = d_z; a.batch = batch; a.n = n;
  a.eps1 = eps1; a.eps2 = eps2;
  const size_t sums_bytes = fx_block_words(6, d) * sizeof(double);
  a.demb_part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + sums_bytes);
  if (!zeroed && hipMemsetAsync(workspace, 0, sums_bytes, st) != hipSuccess) return GDN_ERR_LAUNCH;
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
  if (defer_finish) return gdn_launch_status();     // the caller runs gdn_train_finish after its project backward
  const int total = n * d > d ? n * d : d;
  const int chunks = (n + 256 / (d / 4) - 1) / (256 / (d / 4));
  hipLaunchKernelGGL(gdn_head_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace,
                     a.demb_part, head_parts(batch, chunks, H_BWD1), n, d, d_bn1_w, d_bn1_b, d_bn2_w, d_bn2_b, d_lin_w, d_lin_b, d_emb,
                     zeroed ? const_cast<double*>(stats) : nullptr);
  return gdn_launch_status();
}

extern "C" int gdn_head_train_bwd(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                                  const float* bn1_b, const float* bn2_w, const float* bn2_b,
                                  const float* lin_w, const float* mask, const uint8_t* keep, float keep_scale,
                                  const double* stats, int batch,
                                  int n, int d, float eps1, float eps2, double* workspace, float* d_z,
                                  float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w,
                                  float* d_bn2_b, float* d_lin_w, float* d_lin_b, void* stream) {
  return head_train_bwd_impl(d_out, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, mask, keep, keep_scale, nullptr, 0.f,
                             stats, batch, n, d, eps1, eps2, workspace, d_z, d_emb, d_bn1_w, d_bn1_b, d_bn2_w,
                             d_bn2_b, d_lin_w, d_lin_b, stream);
}

// gdn_head_train_fwd_rng + gdn_mse_loss_grad in the same launches: the last forward pass also writes
// d_out = 2 (out - y) / (batch n) and loss[0] = mean((out - y)^2).  mse_workspace: gdn_head_mse_workspace_bytes(),
// zero-filled once (the ticket is handed back at 0).
extern "C" long long gdn_head_mse_workspace_bytes(void) { return (4096 + 1) * (long long)sizeof(double); }

extern "C" int gdn_head_train_fwd_rng_mse(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                          const float* bn2_w, const float* bn2_b, const float* lin_w,
                                          const float* lin_b, const long long* rng_seed_step, float p_drop,
                                          int batch, int n, int d, float eps1, float eps2, float momentum1,
                                          float momentum2, float* running_mean1, float* running_var1,
                                          long long* batches1, float* running_mean2, float* running_var2,
                                          long long* batches2, double* stats, float* out, const float* y,
                                          double* mse_workspace, float* loss, float* d_out, int buffers_zeroed,
                                          void* stream) {
  if (!rng_seed_step || p_drop < 0.f || p_drop >= 1.f || !y || !mse_workspace || !loss || !d_out) return GDN_ERR_ARG;
  return head_train_fwd_impl(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, lin_b, nullptr, nullptr, 1.f, rng_seed_step,
                             p_drop, batch, n, d, eps1, eps2, momentum1, momentum2, running_mean1, running_var1,
                             batches1, running_mean2, running_var2, batches2, stats, out, stream, nullptr,
                             buffers_zeroed != 0, y, d_out, mse_workspace, loss);
}

extern "C" int gdn_head_train_bwd_rng(const float* d_out, const float* z, const float* emb, const float* bn1_w,
                                      const float* bn1_b, const float* bn2_w, const float* bn2_b,
                                      const float* lin_w, const long long* rng_seed_step, float p_drop,
                                      const double* stats, int batch, int n, int d, float eps1, float eps2,
                                      double* workspace, float* d_z, float* d_emb, float* d_bn1_w, float* d_bn1_b,
                                      float* d_bn2_w, float* d_bn2_b, float* d_lin_w, float* d_lin_b,
                                      int buffers_zeroed, void* stream) {
  if (!rng_seed_step || p_drop < 0.f || p_drop >= 1.f) return GDN_ERR_ARG;
  return head_train_bwd_impl(d_out, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, nullptr, nullptr, 1.f, rng_seed_step,
                             p_drop, stats, batch, n, d, eps1, eps2, workspace, d_z, d_emb, d_bn1_w, d_bn1_b,
                             d_bn2_w, d_bn2_b, d_lin_w, d_lin_b, stream, nullptr, (buffers_zeroed & 1) != 0,
                             (buffers_zeroed & 2) != 0);
}

// MLP head (out_layer_num > 1, models/GDN.py:27-56): the same passes, ending at the [BN, d] activation
// after dropout (forward) / starting from its gradient (backward) instead of the fused Linear(d -> 1).
extern "C" int gdn_head_train_fwd_act(const float* z, const float* emb, const float* bn1_w, const float* bn1_b,
                                      const float* bn2_w, const float* bn2_b, const float* mask,
                                      const uint8_t* keep, float keep_scale, const long long* rng_seed_step,
                                      float p_drop, int batch, int n, int d, float eps1,
                                      float eps2, float momentum1, float momentum2, float* running_mean1,
                                      float* running_var1, long long* batches1, float* running_mean2,
                                      float* running_var2, long long* batches2, double* stats, float* act,
                                      int buffers_zeroed, void* stream) {
  if (!act || p_drop < 0.f || p_drop >= 1.f) return GDN_ERR_ARG;
  return head_train_fwd_impl(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, nullptr, nullptr, mask, keep, keep_scale,
                             rng_seed_step, p_drop, batch, n, d, eps1, eps2, momentum1, momentum2, running_mean1,
                             running_var1, batches1, running_mean2, running_var2, batches2, stats, nullptr, stream,
                             act, buffers_zeroed != 0);
}

extern "C" int gdn_head_train_bwd_act(const float* d_act, const float* z, const float* emb, const float* bn1_w,
                                      const float* bn1_b, const float* bn2_w, const float* bn2_b,
                                      const float* mask, const uint8_t* keep, float keep_scale,
                                      const long long* rng_seed_step, float p_drop,
                                      const double* stats, int batch, int n, int d, float eps1, float eps2,
                                      double* workspace, float* d_z, float* d_emb, float* d_bn1_w, float* d_bn1_b,
                                      float* d_bn2_w, float* d_bn2_b, int buffers_zeroed, void* stream) {
  if (!d_act || p_drop < 0.f || p_drop >= 1.f) return GDN_ERR_ARG;
  return head_train_bwd_impl(nullptr, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, nullptr, mask, keep, keep_scale,
                             rng_seed_step, p_drop, stats, batch, n, d, eps1, eps2, workspace, d_z, d_emb, d_bn1_w,
                             d_bn1_b, d_bn2_w, d_bn2_b, nullptr, nullptr, stream, d_act, buffers_zeroed != 0);
}

// gdn_head_train_bwd_rng called with buffers_zeroed | 2 leaves out its finishing launch; gdn_project_bwd_partials
// leaves out the reduce launch; this runs both in one (see gdn_train_tail_kernel).  head_workspace / stats /
// gradient pointers: as passed to gdn_head_train_bwd_rng; proj_workspace / proj_rows: from
// gdn_project_bwd_partials.  stats is cleared together with the workspace sums when head_zeroed != 0.
extern "C" int gdn_train_finish(double* head_workspace, double* stats, int head_zeroed, int batch, int n, int d,
                                float* d_emb, float* d_bn1_w, float* d_bn1_b, float* d_bn2_w, float* d_bn2_b,
                                float* d_lin_w, float* d_lin_b, const float* proj_workspace, int proj_rows, int w,
                                float* d_proj_w, float* d_a, float* d_c, void* stream) {
  if (!head_workspace || !stats || !d_emb || !d_bn1_w || !d_bn1_b || !d_bn2_w || !d_bn2_b || !d_lin_w || !d_lin_b ||
      !proj_workspace || !d_proj_w || !d_a || !d_c || batch <= 0 || n <= 0 || w <= 0 || proj_rows <= 0)
    return GDN_ERR_ARG;
  if (d != 16 && d != 32 && d != 64 && d != 128) return GDN_ERR_UNSUPPORTED;
  TailArgs t = {};
  const size_t sums_bytes = fx_block_words(6, d) * sizeof(double);
  const int chunks = (n + 256 / (d / 4) - 1) / (256 / (d / 4));
  const int total = n * d > d ? n * d : d;
  t.head_ws = head_workspace;
  t.demb_part = reinterpret_cast<const float*>(reinterpret_cast<const char*>(head_workspace) + sums_bytes);
  t.parts = head_parts(batch, chunks, H_BWD1); t.n = n; t.d = d;
  t.d_bn1_w = d_bn1_w; t.d_bn1_b = d_bn1_b; t.d_bn2_w = d_bn2_w; t.d_bn2_b = d_bn2_b;
  t.d_lin_w = d_lin_w; t.d_lin_b = d_lin_b; t.d_emb = d_emb;
  t.zero_stats = head_zeroed ? stats : nullptr;
  t.g1 = (total + 255) / 256;
  t.w = w; t.wp = w <= 8 ? 8 : ((w + 15) & ~15);
  t.proj_part = proj_workspace; t.rows = proj_rows; t.d_proj_w = d_proj_w; t.d_a = d_a; t.d_c = d_c;
  const int len = d * t.wp + 128 + 2 * n;
  hipLaunchKernelGGL(gdn_train_tail_kernel, dim3(t.g1 + (len + 15) / 16), dim3(256), 0, (hipStream_t)stream, t);
  return gdn_launch_status();
}

// ---- Adam over ONE flat parameter buffer (reference train.py:31,73: torch.optim.Adam(lr, weight_decay)) ----
// p, g, m, v: flat fp32 [count] (every parameter of the model back to back; the gradients are written
// straight into g by the backward kernels, and with several ranks g is what the single all-reduce sums:
// grad_scale = 1 / ranks turns the sum into the mean).  step[0] = number of steps taken so far, in device
// memory, incremented here — it is also the counter of the in-kernel dropout draw.  The update is
// torch.optim.Adam's (amsgrad off, maximize off), in its single-tensor order of operations:
//   g += wd p;  m = m + (g - m)(1 - b1);  v = b2 v + (1 - b2) g g;
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One workgroup (the state is a few 10^4 values): its thread 0 alone touches the counter.
__global__ __launch_bounds__(512) void gdn_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        long long* __restrict__ step, int count, double lr_d,
                                                        double beta1_d, double beta2_d, double eps_d, double wd_d,
                                                        double grad_scale_d, int zero_from, int zero_count) {
  // scalars arrive as the Python floats torch's optimizer holds (double) and are rounded to fp32 where torch
  // rounds them: 1 - beta in double first (1 - 0.999f is 4.7e-5 off 0.001)
  const long long t = step[0] + 1;
  __shared__ float s_bias[2];
  // float4 chunks, ALL of a thread's loads issued before the bias-correction scalars are needed (the two
  // double-precision pow() of thread 0 run under that latency); buffers are 16-byte aligned, count % 4 == 0
  // is handled by a scalar tail
  constexpr int CH = 10;                                 // chunks per thread per round (512 threads: 20480 values, 160 VGPRs)
  const float beta2 = (float)beta2_d, omb1 = (float)(1.0 - beta1_d), omb2 = (float)(1.0 - beta2_d);
  const float eps = (float)eps_d, wd = (float)wd_d, grad_scale = (float)grad_scale_d;
  const int n4 = count >> 2;
  bool have_bias = false;
  float step_size = 0.f, bc2_sqrt = 1.f;
  auto bias = [&]() {
    if (have_bias) return;
    if (threadIdx.x == 0) {      // the two double-precision pow() once, not 1024 times
      // beta^t by squaring in double (t is an integer step count): ~20 multiplies instead of two library
      // pow() calls; agrees with pow() to ~1e-15 relative, far below the fp32 rounding that follows
      auto ipow = [](double b, long long e) {
        double r = 1.0;
        while (e > 0) {
          if (e & 1) r *= b;
          b *= b;
          e >>= 1;
        }
        return r;
      };
      const double bc1 = 1.0 - ipow(beta1_d, t);
      const double bc2 = 1.0 - ipow(beta2_d, t);
      s_bias[0] = (float)(lr_d / bc1);
      s_bias[1] = (float)sqrt(bc2);
    }
    __syncthreads();
    step_size = s_bias[0];
    bc2_sqrt = s_bias[1];
    have_bias = true;
  };
  auto update = [&](float pi, float graw, float mi, float vi_in, int i, float& po, float& mo, float& vo, float& go) {
    float gi = graw * grad_scale;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    mo = mi + (gi - mi) * omb1;
    vo = fmaf(vi_in, beta2, omb2 * gi * gi);
    const float denom = sqrtf(vo) / bc2_sqrt + eps;
    po = pi - step_size * (mo / denom);
    // gradient slots the next backward ACCUMULATES into (atomics) are handed back cleared
    go = (i >= zero_from && i < zero_from + zero_count) ? 0.f : graw;
  };
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int base = 0; base < n4; base += CH * (int)blockDim.x) {
    float4 pp[CH], gg[CH], mm[CH], vv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int i4 = base + c * (int)blockDim.x + (int)threadIdx.x;
      if (i4 < n4) { pp[c] = p4[i4]; gg[c] = g4[i4]; mm[c] = m4[i4]; vv[c] = v4[i4]; }
    }
    bias();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int i4 = base + c * (int)blockDim.x + (int)threadIdx.x;
      if (i4 < n4) {
        float4 po, mo, vo, go;
        update(pp[c].x, gg[c].x, mm[c].x, vv[c].x, 4 * i4 + 0, po.x, mo.x, vo.x, go.x);
        update(pp[c].y, gg[c].y, mm[c].y, vv[c].y, 4 * i4 + 1, po.y, mo.y, vo.y, go.y);
        update(pp[c].z, gg[c].z, mm[c].z, vv[c].z, 4 * i4 + 2, po.z, mo.z, vo.z, go.z);
        update(pp[c].w, gg[c].w, mm[c].w, vv[c].w, 4 * i4 + 3, po.w, mo.w, vo.w, go.w);
        p4[i4] = po; m4[i4] = mo; v4[i4] = vo;
        if (4 * i4 + 3 >= zero_from && 4 * i4 < zero_from + zero_count) g4[i4] = go;
      }
    }
  }
  bias();
  for (int i = 4 * n4 + (int)threadIdx.x; i < count; i += blockDim.x) {
    float po, mo, vo, go;
    update(p[i], g[i], m[i], v[i], i, po, mo, vo, go);
    p[i] = po; m[i] = mo; v[i] = vo;
    if (i >= zero_from && i < zero_from + zero_count) g[i] = go;
  }
  __syncthreads();
  if (threadIdx.x == 0) step[0] = t;
}

extern "C" int gdn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                             long long* step, int count, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double grad_scale, int zero_from, int zero_count, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !step || count <= 0 || zero_from < 0 || zero_count < 0)
    return GDN_ERR_ARG;
  hipLaunchKernelGGL(gdn_adam_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, step, count, lr, beta1, beta2, eps, weight_decay, grad_scale, zero_from,
                     zero_count);
  return gdn_launch_status();
}
