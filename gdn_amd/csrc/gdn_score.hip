// Anomaly scoring on device, float64 like the reference (evaluate.py:48-68, util/data.py:75-82,
// evaluate.py:131-139).  numpy evaluates  a + d*t  and  (x - m) / (|r| + eps)  as separate
// roundings, so FMA contraction is switched off for this file.
#pragma clang fp contract(off)
#include "gdn_common.hpp"

#include <stdlib.h>

namespace {

constexpr int NQ = 6;  // order statistics per sensor: median lo/hi, q25 lo/hi, q75 lo/hi

struct SelectArgs {
  int rank[NQ];      // 0-based ranks in the ascending order of |pred-gt|
  double gamma[2];   // interpolation weights of the 25th / 75th percentile (numpy 'linear')
  int median_pair;   // 1: t even, median = mean of two middle values
};

// delta[s][tick] = |pred[tick][s] - gt[tick][s]| in float64, transposed through LDS so that both
// the fp32 reads (along sensors) and the fp64 writes (along ticks) are coalesced.
__global__ __launch_bounds__(256) void score_delta_kernel(const float* __restrict__ pred,
                                                          const float* __restrict__ gt, int t, int n,
                                                          double* __restrict__ ws) {
  __shared__ double tile[64][65];
  const int t0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = wv; r < 64; r += 4) {
    const int tick = t0 + r, s = s0 + lane;
    if (tick < t && s < n) {
      const size_t o = (size_t)tick * n + s;
      tile[r][lane] = fabs((double)pred[o] - (double)gt[o]);
    }
  }
  __syncthreads();
  for (int r = wv; r < 64; r += 4) {
    const int s = s0 + r, tick = t0 + lane;
    if (s < n && tick < t) ws[(size_t)s * t + tick] = tile[lane][r];
  }
}

// ------------------------------------------------------------------ radix select
// Exact order statistics by an MSB-first 8-bit radix select of NQ ranks per sensor over the sensor's t
// keys (non-negative doubles order like their bit patterns), spread over the whole chip: one launch per
// digit, grid =
// (key slices of 2048, sensors).  A block keeps its <= 2048 keys in registers (8 per thread),
// histograms the current digit of the keys that still match some rank's prefix in LDS, adds its
// non-empty bins to the sensor's global histogram, and takes a ticket; the LAST block of a sensor
// (ticket == slices-1, no spinning anywhere) locates every rank's bin and extends the prefixes for
// the next launch.  From the third digit on, the matching keys are compacted into a second buffer,
// so later launches read a few hundred keys per sensor instead of all t.
constexpr int SLICE = 2048;

struct SelState {
  unsigned long long prefix[NQ];
  int rem[NQ];
  int rep[NQ];
  unsigned int hist[NQ][256];
  unsigned int arrive;
  unsigned int cnt_in;    // keys the next pass reads
  unsigned int cnt_out;   // survivors written by the running pass
  unsigned int src;       // buffer the next pass reads: 0 = A, 1 = B
};

__global__ void select_init_kernel(SelState* __restrict__ state, int n, int t, const SelectArgs sa) {
  const int s = blockIdx.x;
  SelState& st = state[s];
  for (int i = threadIdx.x; i < NQ * 256; i += blockDim.x) (&st.hist[0][0])[i] = 0u;
  if (threadIdx.x < NQ) {
    st.prefix[threadIdx.x] = 0ull;
    st.rem[threadIdx.x] = sa.rank[threadIdx.x];
    st.rep[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0) {
    st.arrive = 0u;
    st.cnt_in = (unsigned int)t;
    st.cnt_out = 0u;
    st.src = 0u;
  }
}

template <bool COMPACT>
__global__ __launch_bounds__(256) void select_pass_kernel(unsigned long long* __restrict__ bufA,
                                                          unsigned long long* __restrict__ bufB,
                                                          SelState* __restrict__ state, int t, int pass,
                                                          const SelectArgs sa, double* __restrict__ med_iqr) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long stage[COMPACT ? SLICE : 1];
  __shared__ unsigned int n_stage, out_base, ticket_s;
  const int s = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SelState& st = state[s];
  const unsigned int cnt = st.cnt_in;
  const unsigned long long* in = (st.src ? bufB : bufA) + (size_t)s * t;
  unsigned long long* out = (st.src ? bufA : bufB) + (size_t)s * t;
  const int shift = 56 - 8 * pass;
  unsigned long long pf[NQ];
  bool active[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    pf[q] = st.prefix[q];
    active[q] = st.rep[q] == q;
  }
  for (int i = tid; i < NQ * 256; i += 256) (&hist[0][0])[i] = 0u;
  if (tid == 0) n_stage = 0u;
  __syncthreads();

  // a block walks slices g, g+G, ... of the input (late passes are launched with few blocks per
  // sensor because compaction normally leaves a handful of keys; any count stays correct)
  for (unsigned int base = (unsigned int)g * SLICE; base < cnt; base += (unsigned int)G * SLICE) {
    unsigned long long key[8];
    bool have[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned int i = base + tid + u * 256;
      have[u] = i < cnt;
      key[u] = in[min(i, cnt - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned int digit = (unsigned int)(key[u] >> shift) & 255u;
      const unsigned int d0 = __builtin_amdgcn_readfirstlane(digit);
      bool any = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!active[q]) continue;   // uniform
        const bool match = have[u] && (pass == 0 ? true : ((key[u] ^ pf[q]) >> (shift + 8)) == 0ull);
        any |= match;
        if (__all(match && digit == d0)) {
          if (lane == 0) atomicAdd(&hist[q][d0], 64u);
        } else if (match) {
          atomicAdd(&hist[q][digit], 1u);
        }
      }
      if constexpr (COMPACT) {
        if (any) stage[atomicAdd(&n_stage, 1u)] = key[u];
      }
    }
    if constexpr (COMPACT) {
      // flush this slice's survivors before the staging area is reused
      __syncthreads();
      if (tid == 0) out_base = n_stage ? atomicAdd(&st.cnt_out, n_stage) : 0u;
      __syncthreads();
      for (unsigned int i = tid; i < n_stage; i += 256) out[out_base + i] = stage[i];
      __syncthreads();
      if (tid == 0) n_stage = 0u;
      __syncthreads();
    }
  }
  __syncthreads();
  // contribute: non-empty bins -> the sensor's histogram; survivors -> the other buffer
  for (int i = tid; i < NQ * 256; i += 256) {
    const unsigned int c = (&hist[0][0])[i];
    if (c) atomicAdd(&(&st.hist[0][0])[i], c);
  }
  // Hand-off without fences: every contribution the last block consumes (histogram bins, survivor
  // count) is an agent-scope ATOMIC, performed at the device coherence point; each wave waits for
  // its own atomics to be acknowledged, the barrier orders all waves before the ticket, and the last
  // block reads the totals back with atomic exchanges (which also re-zero them).  The compacted keys
  // are plain stores: they are only read by the NEXT launch.  (__threadfence() here costs a full L2
  // write-back per block: 150 us per pass with 2000 blocks.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket_s = atomicAdd(&st.arrive, 1u);
  __syncthreads();
  if (ticket_s != (unsigned int)(G - 1)) return;

  // ---- last block of this sensor
  for (int i = tid; i < NQ * 256; i += 256) (&hist[0][0])[i] = atomicExch(&(&st.hist[0][0])[i], 0u);
  __syncthreads();
  __shared__ unsigned long long npf[NQ];
  __shared__ int nrem[NQ];
  for (int q = wv; q < NQ; q += 4) {
    const unsigned int* h = hist[st.rep[q]];
    const uint4 c4 = *reinterpret_cast<const uint4*>(h + 4 * lane);
    const int mine = (int)(c4.x + c4.y + c4.z + c4.w);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const int left0 = st.rem[q];
    const int excl = incl - mine;
    if (left0 >= excl && left0 < incl) {
      int left = left0 - excl, bin = 0;
      const int c[4] = {(int)c4.x, (int)c4.y, (int)c4.z, (int)c4.w};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (bin == j && left >= c[j]) {
          left -= c[j];
          bin = j + 1;
        }
      }
      npf[q] = pf[q] | ((unsigned long long)(4 * lane + bin) << shift);
      nrem[q] = left;
    }
  }
  __syncthreads();
  if (tid < NQ) {
    st.prefix[tid] = npf[tid];
    st.rem[tid] = nrem[tid];
    int r = tid;
    for (int q = tid - 1; q >= 0; --q)
      if (npf[q] == npf[tid]) r = q;
    st.rep[tid] = r;
  }
  if (tid == 0) {
    atomicExch(&st.arrive, 0u);
    if (COMPACT) {
      st.cnt_in = atomicExch(&st.cnt_out, 0u);
      st.src ^= 1u;
    }
    if (pass == 7) {
      double v[NQ];
      for (int q = 0; q < NQ; ++q) v[q] = __longlong_as_double((long long)npf[q]);
      const double med = sa.median_pair ? (v[0] + v[1]) / 2.0 : v[0];
      double qv[2];
      for (int h = 0; h < 2; ++h) {   // numpy _lerp
        const double a = v[2 + 2 * h], b = v[3 + 2 * h], gm = sa.gamma[h];
        const double diff = b - a;
        double r = a + diff * gm;
        if (gm >= 0.5) r = b - diff * (1.0 - gm);
        qv[h] = r;
      }
      med_iqr[2 * s] = med;
      med_iqr[2 * s + 1] = qv[1] - qv[0];
    }
  }
}

// Finisher: digits first_pass..7 of every rank in ONE launch, one block per sensor, no global
// hand-offs.  Runs after the compacting passes, when a sensor normally has a handful of keys left
// (<= 2048 stay in registers; more — e.g. thousands of identical values — are re-read per digit).
__global__ __launch_bounds__(256) void select_finish_kernel(const unsigned long long* __restrict__ bufA,
                                                            const unsigned long long* __restrict__ bufB,
                                                            SelState* __restrict__ state, int t, int first_pass,
                                                            const SelectArgs sa, double* __restrict__ med_iqr) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long prefix[NQ];
  __shared__ int rem[NQ];
  __shared__ int rep[NQ];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SelState& st = state[s];
  const unsigned int cnt = st.cnt_in;
  const unsigned long long* in = (st.src ? bufB : bufA) + (size_t)s * t;
  if (tid < NQ) {
    prefix[tid] = st.prefix[tid];
    rem[tid] = st.rem[tid];
  }
  const bool resident = cnt <= SLICE;
  unsigned long long key[8];
  if (resident) {
#pragma unroll
    for (int u = 0; u < 8; ++u) key[u] = in[min((unsigned int)(tid + u * 256), cnt - 1)];
  }
  __syncthreads();
  for (int pass = first_pass; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < NQ) {
      int r = tid;
      for (int q = tid - 1; q >= 0; --q)
        if (prefix[q] == prefix[tid]) r = q;
      rep[tid] = r;
    }
    for (int i = tid; i < NQ * 256; i += 256) (&hist[0][0])[i] = 0u;
    __syncthreads();
    unsigned long long pf[NQ];
    bool active[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pf[q] = prefix[q];
      active[q] = rep[q] == q;
    }
    auto tally = [&](unsigned long long k, bool have) {
      const unsigned int digit = (unsigned int)(k >> shift) & 255u;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!active[q]) continue;
        if (have && (pass == 0 || ((k ^ pf[q]) >> (shift + 8)) == 0ull)) atomicAdd(&hist[q][digit], 1u);
      }
    };
    if (resident) {
#pragma unroll
      for (int u = 0; u < 8; ++u) tally(key[u], (unsigned int)(tid + u * 256) < cnt);
    } else {
      for (unsigned int i = tid; i < cnt; i += 256) tally(in[i], true);
    }
    __syncthreads();
    for (int q = wv; q < NQ; q += 4) {
      const unsigned int* h = hist[rep[q]];
      const uint4 c4 = *reinterpret_cast<const uint4*>(h + 4 * lane);
      const int mine = (int)(c4.x + c4.y + c4.z + c4.w);
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const int left0 = rem[q];
      const int excl = incl - mine;
      if (left0 >= excl && left0 < incl) {
        int left = left0 - excl, bin = 0;
        const int c[4] = {(int)c4.x, (int)c4.y, (int)c4.z, (int)c4.w};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (bin == j && left >= c[j]) {
            left -= c[j];
            bin = j + 1;
          }
        }
        prefix[q] |= (unsigned long long)(4 * lane + bin) << shift;
        rem[q] = left;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    double v[NQ];
    for (int q = 0; q < NQ; ++q) v[q] = __longlong_as_double((long long)prefix[q]);
    const double med = sa.median_pair ? (v[0] + v[1]) / 2.0 : v[0];
    double qv[2];
    for (int h = 0; h < 2; ++h) {   // numpy _lerp
      const double a = v[2 + 2 * h], b = v[3 + 2 * h], gm = sa.gamma[h];
      const double diff = b - a;
      double r = a + diff * gm;
      if (gm >= 0.5) r = b - diff * (1.0 - gm);
      qv[h] = r;
    }
    med_iqr[2 * s] = med;
    med_iqr[2 * s + 1] = qv[1] - qv[0];
  }
}

// Normalise, smooth, max.  One wave owns a run of consecutive ticks; lane l owns sensors l, l+64, ...
// and slides a 4-deep window of normalised errors down the run, so each (tick, sensor) value — and
// its float64 division — is computed once.  smoothed = mean of the value at the tick and its 3
// predecessors (0 for the first 3 ticks of the SERIES), anomaly = max over sensors (wave reduction).
// Rows before this shard's first tick come from the halo [3][n].
constexpr int RUN = 8;        // ticks per wave (short runs: thousands of waves keep loads in flight)

__global__ __launch_bounds__(256) void score_smooth_max_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, const double* __restrict__ med_iqr,
    int t, int n, int first_tick, const float* __restrict__ halo_pred, const float* __restrict__ halo_gt,
    double* __restrict__ scores, double* __restrict__ anomaly) {
  __shared__ double best[4][RUN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int nruns = (t + RUN - 1) / RUN;
  for (int run = blockIdx.x * wpb + wv; run < nruns; run += gridDim.x * wpb) {
    const int t0 = run * RUN, t1 = min(t, t0 + RUN);
    for (int s0 = 0; s0 < n; s0 += 128) {          // two sensors per lane per sweep
      const int sa_ = s0 + lane, sb_ = s0 + 64 + lane;
      const bool la = sa_ < n, lb = sb_ < n;
      const int ca = la ? sa_ : n - 1, cb = lb ? sb_ : n - 1;
      const double meda = med_iqr[2 * ca], dena = fabs(med_iqr[2 * ca + 1]) + 1e-2;
      const double medb = med_iqr[2 * cb], denb = fabs(med_iqr[2 * cb + 1]) + 1e-2;
      auto norm = [&](int tt, int sc, double med, double den) -> double {   // tt < 0: halo row
        const float* pp = tt >= 0 ? pred + (size_t)tt * n : halo_pred + (size_t)(3 + tt) * n;
        const float* gg = tt >= 0 ? gt + (size_t)tt * n : halo_gt + (size_t)(3 + tt) * n;
        return (fabs((double)pp[sc] - (double)gg[sc]) - med) / den;
      };
      // the 3 predecessors of t0 that exist in the series (a missing one only feeds ticks whose
      // series index is < 3, which are forced to 0)
      const int g0 = first_tick + t0;
      double a3 = g0 >= 3 ? norm(t0 - 3, ca, meda, dena) : 0.0;
      double a2 = g0 >= 2 ? norm(t0 - 2, ca, meda, dena) : 0.0;
      double a1 = g0 >= 1 ? norm(t0 - 1, ca, meda, dena) : 0.0;
      double b3 = g0 >= 3 ? norm(t0 - 3, cb, medb, denb) : 0.0;
      double b2 = g0 >= 2 ? norm(t0 - 2, cb, medb, denb) : 0.0;
      double b1 = g0 >= 1 ? norm(t0 - 1, cb, medb, denb) : 0.0;
#pragma unroll 8
      for (int tick = t0; tick < t1; ++tick) {
        const double a0 = norm(tick, ca, meda, dena);
        const double b0 = norm(tick, cb, medb, denb);
        double sma = 0.0, smb = 0.0;
        if (first_tick + tick >= 3) {               // numpy sums the 4 values left to right
          sma = (((a3 + a2) + a1) + a0) / 4.0;
          smb = (((b3 + b2) + b1) + b0) / 4.0;
        }
        if (scores) {
          if (la) scores[(size_t)sa_ * t + tick] = sma;
          if (lb) scores[(size_t)sb_ * t + tick] = smb;
        }
        double m = fmax(la ? sma : -INFINITY, lb ? smb : -INFINITY);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d));
        if (lane == 0) best[wv][tick - t0] = s0 == 0 ? m : fmax(best[wv][tick - t0], m);
        a3 = a2; a2 = a1; a1 = a0;
        b3 = b2; b2 = b1; b1 = b0;
      }
    }
    // (LDS writes above and reads below are by the same wave, in program order)
    if (lane < t1 - t0) anomaly[t0 + lane] = best[wv][lane];
  }
}

}  // namespace

// workspace: two key buffers [n,t] (u64 bit patterns of |pred-gt|) + one SelState per sensor
extern "C" long long gdn_score_workspace_bytes(int t, int n) {
  if (t <= 0 || n <= 0) return 0;
  return 2ll * t * n * 8 + (long long)n * (long long)sizeof(SelState) + 256;
}

extern "C" int gdn_score_quantiles(const float* pred, const float* gt, int t, int n, double* workspace,
                                   double* med_iqr, void* stream) {
  if (!pred || !gt || !workspace || !med_iqr || t <= 0 || n <= 0) return GDN_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(score_delta_kernel, dim3((t + 63) / 64, (n + 63) / 64), dim3(256), 0, st, pred, gt, t,
                     n, workspace);
  SelectArgs sa;
  // np.median: middle value, or the mean of the two middle values when t is even
  sa.median_pair = (t % 2 == 0);
  sa.rank[0] = sa.median_pair ? t / 2 - 1 : (t - 1) / 2;
  sa.rank[1] = sa.median_pair ? t / 2 : (t - 1) / 2;
  // np.percentile(method='linear'): virtual index n*q + (alpha + q*(1-alpha-beta)) - 1, alpha=beta=1
  const double qs[2] = {25.0 / 100.0, 75.0 / 100.0};
  for (int h = 0; h < 2; ++h) {
    const double vi = (double)t * qs[h] + (1.0 + qs[h] * (1.0 - 1.0 - 1.0)) - 1.0;
    double lo = floor(vi);
    int ilo = (int)lo, ihi = ilo + 1;
    if (ilo < 0) ilo = 0;
    if (ihi > t - 1) ihi = t - 1;
    if (ilo > t - 1) ilo = t - 1;
    sa.rank[2 + 2 * h] = ilo;
    sa.rank[3 + 2 * h] = ihi;
    sa.gamma[h] = vi - lo;
  }
  unsigned long long* bufA = reinterpret_cast<unsigned long long*>(workspace);
  unsigned long long* bufB = bufA + (size_t)t * n;
  SelState* state = reinterpret_cast<SelState*>(bufB + (size_t)t * n);
  const int slices = (t + SLICE - 1) / SLICE;
  hipLaunchKernelGGL(select_init_kernel, dim3(n), dim3(256), 0, st, state, n, t, sa);
  // digits 0-1 (exponent bytes: no point compacting), 2-3 compacting (3 normally sees a few
  // thousand keys per sensor: two blocks each), then one finisher launch for digits 4-7
  const int wide = slices > 1 ? 4 : 0;   // a single slice goes straight to the finisher
  for (int pass = 0; pass < wide; ++pass) {
    if (pass < 2)
      hipLaunchKernelGGL(select_pass_kernel<false>, dim3(slices, n), dim3(256), 0, st, bufA, bufB, state, t,
                         pass, sa, med_iqr);
    else
      hipLaunchKernelGGL(select_pass_kernel<true>, dim3(pass == 2 ? slices : min(slices, 2), n), dim3(256), 0,
                         st, bufA, bufB, state, t, pass, sa, med_iqr);
  }
  hipLaunchKernelGGL(select_finish_kernel, dim3(n), dim3(256), 0, st, bufA, bufB, state, t, wide, sa, med_iqr);
  return gdn_launch_status();
}

extern "C" int gdn_score_smooth_max(const float* pred, const float* gt, const double* med_iqr, int t, int n,
                                    int first_tick, const float* halo_pred, const float* halo_gt,
                                    double* scores, double* anomaly, void* stream) {
  if (!pred || !gt || !med_iqr || !anomaly || t <= 0 || n <= 0 || first_tick < 0) return GDN_ERR_ARG;
  if (first_tick > 0 && (!halo_pred || !halo_gt)) return GDN_ERR_ARG;
  const int runs = (t + RUN - 1) / RUN;
  const int grid = min((runs + 3) / 4, gdn_cu_count() * 8);
  hipLaunchKernelGGL(score_smooth_max_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, gt,
                     med_iqr, t, n, first_tick, halo_pred, halo_gt, scores, anomaly);
  return gdn_launch_status();
}
