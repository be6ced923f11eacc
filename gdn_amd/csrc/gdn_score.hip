// Anomaly scoring on device, float64 like the reference (evaluate.py:48-68, util/data.py:75-82,
// evaluate.py:131-139).  numpy evaluates  a + d*t  and  (x - m) / (|r| + eps)  as separate
// roundings, so FMA contraction is switched off for this file.
#pragma clang fp contract(off)
#include "gdn_common.hpp"

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

// One workgroup per sensor: MSB-first radix select (8-bit digits) of NQ ranks at once over the
// sensor's t keys (non-negative doubles order like their bit patterns).  Ranks whose prefixes
// still coincide share one histogram.
//  * KPT > 0: every thread keeps KPT keys in registers for all 8 passes (t <= KPT * blockDim) plus
//    an ALIVE bit per key: a key that matches no rank's prefix can never match again, so after
//    the first two or three digits almost every wave skips its keys and the late passes are free;
//    KPT == 0: keys are re-read from memory each pass (any t).
//  * a whole wave landing in one bin (the exponent bytes) does one add of 64;
//  * the bin holding each rank is found by one wave per rank with a shuffle scan (not a serial
//    walk over 256 LDS words).
template <int KPT>
__global__ __launch_bounds__(1024) void score_select_kernel(const double* __restrict__ ws, int t,
                                                            const SelectArgs sa, double* __restrict__ med_iqr) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long prefix[NQ];
  __shared__ int rem[NQ];
  __shared__ int rep[NQ];
  const int s = blockIdx.x;
  const unsigned long long* keys = reinterpret_cast<const unsigned long long*>(ws) + (size_t)s * t;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6;
  constexpr int NK = KPT > 0 ? KPT : 1;
  unsigned long long kreg[NK];
  unsigned int alive = 0u;
  if constexpr (KPT > 0) {
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
      const int i = tid + u * nth;
      kreg[u] = keys[min(i, t - 1)];                          // unconditional, clamped
      if (i < t) alive |= 1u << u;
    }
  }
  if (tid < NQ) {
    prefix[tid] = 0ull;
    rem[tid] = sa.rank[tid];
  }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < NQ) {
      int r = tid;
      for (int q = tid - 1; q >= 0; --q)
        if (prefix[q] == prefix[tid]) r = q;
      rep[tid] = r;
    }
    for (int i = tid; i < NQ * 256; i += nth) (&hist[0][0])[i] = 0u;
    __syncthreads();
    unsigned long long pf[NQ];
    bool active[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pf[q] = prefix[q];
      active[q] = rep[q] == q;
    }
    // returns whether the key still matches some rank's prefix
    auto tally = [&](unsigned long long key, bool live) -> bool {
      const unsigned int digit = (unsigned int)(key >> shift) & 255u;
      const unsigned int d0 = __builtin_amdgcn_readfirstlane(digit);
      bool any = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!active[q]) continue;   // uniform
        // high bits above this digit must equal the prefix (pass 0: no high bits)
        const bool match = live && (pass == 0 ? true : ((key ^ pf[q]) >> (shift + 8)) == 0ull);
        any |= match;
        if (__all(match && digit == d0)) {
          if (lane == 0) atomicAdd(&hist[q][d0], 64u);   // whole wave in one bin: one add of 64
        } else if (match) {
          atomicAdd(&hist[q][digit], 1u);
        }
      }
      return any;
    };
    if constexpr (KPT > 0) {
#pragma unroll
      for (int u = 0; u < KPT; ++u) {
        const bool live = (alive >> u) & 1u;
        if (__any(live)) {                       // wave-uniform skip of dead keys
          if (!tally(kreg[u], live)) alive &= ~(1u << u);
        }
      }
    } else {
      for (int i0 = 0; i0 < t; i0 += 4 * nth) {
        unsigned long long k4[4];
        bool l4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + tid + u * nth;
          k4[u] = keys[min(i, t - 1)];
          l4[u] = i < t;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) tally(k4[u], l4[u]);
      }
    }
    __syncthreads();
    // wave q locates rank q's bin: lane l owns bins 4l..4l+3, inclusive scan over lanes
    if (wv < NQ) {
      const unsigned int* h = hist[rep[wv]];
      const uint4 c4 = *reinterpret_cast<const uint4*>(h + 4 * lane);
      const int mine = (int)(c4.x + c4.y + c4.z + c4.w);
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const int left0 = rem[wv];
      const int excl = incl - mine;
      // the owning lane is the first whose inclusive count exceeds the remaining rank
      const bool owner = left0 >= excl && left0 < incl;
      if (owner) {
        int left = left0 - excl, bin = 4 * lane;
        const int c[4] = {(int)c4.x, (int)c4.y, (int)c4.z, (int)c4.w};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (left >= c[j] && bin == 4 * lane + j) {
            left -= c[j];
            bin += 1;
          }
        }
        prefix[wv] |= (unsigned long long)bin << shift;
        rem[wv] = left;
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
      const double a = v[2 + 2 * h], b = v[3 + 2 * h], g = sa.gamma[h];
      const double diff = b - a;
      double r = a + diff * g;
      if (g >= 0.5) r = b - diff * (1.0 - g);
      qv[h] = r;
    }
    med_iqr[2 * s] = med;
    med_iqr[2 * s + 1] = qv[1] - qv[0];
  }
}

// One wave per tick: lanes stride over sensors; smoothed score = mean of the normalised error at
// the tick and its 3 predecessors (0 for the first 3 ticks of the series), anomaly = max over
// sensors.  Rows before this shard's first tick come from the optional halo [3][n].
__global__ __launch_bounds__(256) void score_smooth_max_kernel(
    const float* __restrict__ pred, const float* __restrict__ gt, const double* __restrict__ med_iqr,
    int t, int n, int first_tick, const float* __restrict__ halo_pred, const float* __restrict__ halo_gt,
    double* __restrict__ scores, double* __restrict__ anomaly) {
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int tick = blockIdx.x * wpb + (threadIdx.x >> 6); tick < t; tick += gridDim.x * wpb) {
    const bool zero = first_tick + tick < 3;   // evaluate.py:62-65: the first 3 ticks stay 0
    double best = -INFINITY;
    for (int s = lane; s < n; s += 64) {
      double sm = 0.0;
      if (!zero) {
        const double med = med_iqr[2 * s];
        const double den = fabs(med_iqr[2 * s + 1]) + 1e-2;
        double acc = 0.0;
#pragma unroll
        for (int back = 3; back >= 0; --back) {
          const int tt = tick - back;
          float p, g;
          if (tt >= 0) {
            p = pred[(size_t)tt * n + s];
            g = gt[(size_t)tt * n + s];
          } else {
            p = halo_pred[(size_t)(3 + tt) * n + s];
            g = halo_gt[(size_t)(3 + tt) * n + s];
          }
          const double a = (fabs((double)p - (double)g) - med) / den;
          acc = back == 3 ? a : acc + a;   // numpy sums the 4 values left to right
        }
        sm = acc / 4.0;
      }
      if (scores) scores[(size_t)s * t + tick] = sm;
      best = fmax(best, sm);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) best = fmax(best, __shfl_xor(best, m));
    if (lane == 0) anomaly[tick] = best;
  }
}

}  // namespace

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
  if (t <= 8 * 1024)
    hipLaunchKernelGGL(score_select_kernel<8>, dim3(n), dim3(1024), 0, st, workspace, t, sa, med_iqr);
  else if (t <= 32 * 1024)
    hipLaunchKernelGGL(score_select_kernel<32>, dim3(n), dim3(1024), 0, st, workspace, t, sa, med_iqr);
  else
    hipLaunchKernelGGL(score_select_kernel<0>, dim3(n), dim3(1024), 0, st, workspace, t, sa, med_iqr);
  return gdn_launch_status();
}

extern "C" int gdn_score_smooth_max(const float* pred, const float* gt, const double* med_iqr, int t, int n,
                                    int first_tick, const float* halo_pred, const float* halo_gt,
                                    double* scores, double* anomaly, void* stream) {
  if (!pred || !gt || !med_iqr || !anomaly || t <= 0 || n <= 0 || first_tick < 0) return GDN_ERR_ARG;
  if (first_tick > 0 && (!halo_pred || !halo_gt)) return GDN_ERR_ARG;
  const int grid = min((t + 3) / 4, gdn_cu_count() * 8);
  hipLaunchKernelGGL(score_smooth_max_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, gt,
                     med_iqr, t, n, first_tick, halo_pred, halo_gt, scores, anomaly);
  return gdn_launch_status();
}
