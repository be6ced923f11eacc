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

// keys[s][tick] = |pred[tick][s] - gt[tick][s]| in float64 (bit pattern = radix key), transposed
// through LDS so that both the fp32 reads (along sensors) and the fp64 writes (along ticks) are
// coalesced.  Row pitch >= t; slots t..pitch-1 get the FILLER pattern (all ones: larger than every real
// key, never counted).
constexpr unsigned long long FILLER = ~0ull;

__global__ __launch_bounds__(256) void score_keys_kernel(const float* __restrict__ pred,
                                                         const float* __restrict__ gt, int t, int n, int pitch,
                                                         double* __restrict__ keys) {
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
    if (s < n && tick < pitch)
      keys[(size_t)s * pitch + tick] = tick < t ? tile[lane][r] : __longlong_as_double((long long)FILLER);
  }
}

// ------------------------------------------------------------------ radix select
// Exact order statistics by an MSB-first 8-bit radix select of NQ ranks per sensor (non-negative
// doubles order like their bit patterns), spread over the whole chip.
//
// Input layout: keys[blocks][n][pitch] — `blocks` row blocks as they arrive from `blocks` ranks in
// the multi-GPU exchange (one block, pitch = t, on a single GPU); sensor s owns blocks*pitch SLOTS,
// of which `total` hold real keys and the rest the FILLER.  The input is never written.
//
// Digits 0-1 (the exponent bytes) and 2: one launch per digit, grid = (2048-slot slices, sensors);
// a block keeps its slice in registers (8 keys per thread), histograms the digit of the keys that
// still match some rank's prefix in LDS (a wave landing in one bin adds 64 at once), adds its
// non-empty bins to the sensor's global histogram and takes a ticket; the LAST block of a sensor
// (no spinning, no fences: every consumed value is an agent-scope atomic) locates every rank's bin
// and extends the prefixes for the next launch.  The digit-2 launch also compacts the matching keys
// (normally a few percent) into the workspace, and one single-block finisher launch per sensor does
// digits 3-7 on those.
constexpr int SLICE = 2048;

struct SelState {
  unsigned long long prefix[NQ];
  int rem[NQ];
  int rep[NQ];
  unsigned int hist[NQ][256];
  unsigned int arrive;
  unsigned int cnt_b;     // keys compacted into the workspace buffer
  unsigned int digits_done;   // digits the prefixes hold: pass 0 also settles digit 1 when it can (see select_pass_kernel)
  unsigned int pad1;
};

struct KeyLayout {
  const unsigned long long* keys;   // [blocks][n][pitch]
  unsigned long long* bufb;         // [n][blocks*pitch] compacted survivors
  int blocks, n, pitch;
};

__global__ void select_init_kernel(SelState* __restrict__ state, const SelectArgs sa) {
  SelState& st = state[blockIdx.x];
  for (int i = threadIdx.x; i < NQ * 256; i += blockDim.x) (&st.hist[0][0])[i] = 0u;
  if (threadIdx.x < NQ) {
    st.prefix[threadIdx.x] = 0ull;
    st.rem[threadIdx.x] = sa.rank[threadIdx.x];
    st.rep[threadIdx.x] = 0;
  }
  if (threadIdx.x == 0) {
    st.arrive = 0u;
    st.cnt_b = 0u;
    st.digits_done = 0u;
  }
}

// wave q's lanes find the bin of rank q in a 256-bin histogram; returns through npf / nrem
__device__ __forceinline__ void locate_bin(const unsigned int* h, int left0, unsigned long long pf, int shift,
                                           int lane, unsigned long long* npf, int* nrem) {
  const uint4 c4 = *reinterpret_cast<const uint4*>(h + 4 * lane);
  const int mine = (int)(c4.x + c4.y + c4.z + c4.w);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
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
    *npf = pf | ((unsigned long long)(4 * lane + bin) << shift);
    *nrem = left;
  }
}

__device__ __forceinline__ void write_result(const unsigned long long* prefix, const SelectArgs& sa, int s,
                                             double* __restrict__ med_iqr) {
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

// FROM_B: read the compacted workspace buffer (flat) instead of the blocked input.
// COMPACT: also write the matching keys to the workspace buffer.
template <bool FROM_B, bool COMPACT>
__global__ __launch_bounds__(256) void select_pass_kernel(const KeyLayout kl, SelState* __restrict__ state,
                                                          int pass) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long stage[COMPACT ? SLICE : 1];
  __shared__ unsigned int n_stage, out_base, ticket_s;
  __shared__ unsigned long long npf[NQ];
  __shared__ int nrem[NQ];
  const int s = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SelState& st = state[s];
  // Pass 0 usually settles TWO digits: the errors of a sensor almost always share the top byte of their keys (sign
  // + 7 exponent bits), so next to the histogram of byte 0 it histograms byte 1 of the keys whose byte 0 equals that
  // of the sensor's first key; if every quantile's rank falls into that byte-0 bin the last workgroup locates
  // digit 1 as well and the pass-1 launch returns at once (it is still launched: the sequence is captured in
  // graphs).  One streaming pass over the keys less: 14.5 -> ~5 us of an 86 us select at the 8-rank shape.
  if (!FROM_B && !COMPACT && pass == 1 && st.digits_done >= 2u) return;
  const unsigned int slots = (unsigned int)kl.blocks * (unsigned int)kl.pitch;
  const unsigned int cnt = FROM_B ? st.cnt_b : slots;
  unsigned long long* outb = kl.bufb + (size_t)s * slots;
  const int shift = 56 - 8 * pass;
  const bool two = !FROM_B && !COMPACT && pass == 0;                    // uniform
  const unsigned int dref = two ? (unsigned int)(kl.keys[(size_t)s * kl.pitch] >> 56) : 0u;   // block 0, slot 0
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

  // a block walks slices g, g+G, ... (any count stays correct when fewer blocks are launched)
  for (unsigned int base = (unsigned int)g * SLICE; base < cnt; base += (unsigned int)G * SLICE) {
    // slice -> memory: blocked input (a slice never straddles two blocks: pitch % SLICE == 0 when
    // blocks > 1) or the flat workspace buffer
    const unsigned long long* in;
    unsigned int lim;   // valid slots of this slice's row, from `base`
    if (FROM_B) {
      in = outb + base;
      lim = cnt - base;
    } else {
      const unsigned int blk = base / (unsigned int)kl.pitch, off = base - blk * (unsigned int)kl.pitch;
      in = kl.keys + ((size_t)blk * kl.n + s) * kl.pitch + off;
      lim = (unsigned int)kl.pitch - off;
    }
    unsigned long long key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned int i = tid + u * 256;
      key[u] = i < lim ? in[i] : FILLER;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool real = key[u] != FILLER;
      const unsigned int digit = (unsigned int)(key[u] >> shift) & 255u;
      const unsigned int d0 = __builtin_amdgcn_readfirstlane(digit);
      bool any = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!active[q]) continue;   // uniform
        const bool match = real && (pass == 0 ? true : ((key[u] ^ pf[q]) >> (shift + 8)) == 0ull);
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
      if (two && real && digit == dref) atomicAdd(&hist[1][(unsigned int)(key[u] >> 48) & 255u], 1u);
    }
    if constexpr (COMPACT) {
      __syncthreads();
      if (tid == 0) out_base = n_stage ? atomicAdd(&st.cnt_b, n_stage) : 0u;
      __syncthreads();
      for (unsigned int i = tid; i < n_stage; i += 256) outb[out_base + i] = stage[i];
      __syncthreads();
      if (tid == 0) n_stage = 0u;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = tid; i < NQ * 256; i += 256) {
    const unsigned int c = (&hist[0][0])[i];
    if (c) atomicAdd(&(&st.hist[0][0])[i], c);
  }
  // Hand-off without fences: every value the last block consumes (histogram bins, survivor count)
  // is an agent-scope ATOMIC performed at the device coherence point; each wave waits for its own
  // atomics to be acknowledged, the barrier orders all waves before the ticket, and the last block
  // reads the totals back with atomic exchanges (which also re-zero them).  The compacted keys are
  // plain stores: only the NEXT launch reads them.  (A __threadfence() here costs a full L2
  // write-back per block: 150 us per pass with 2000 blocks.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket_s = atomicAdd(&st.arrive, 1u);
  __syncthreads();
  if (ticket_s != (unsigned int)(G - 1)) return;

  // ---- last block of this sensor
  for (int i = tid; i < NQ * 256; i += 256) (&hist[0][0])[i] = atomicExch(&(&st.hist[0][0])[i], 0u);
  __syncthreads();
  for (int q = wv; q < NQ; q += 4) locate_bin(hist[st.rep[q]], st.rem[q], pf[q], shift, lane, &npf[q], &nrem[q]);
  __syncthreads();
  unsigned int done = (unsigned int)pass + 1u;
  if (two) {
    bool inside = dref != 255u;                                  // (255: the filler's byte)
    for (int q = 0; q < NQ; ++q) inside = inside && (unsigned int)(npf[q] >> 56) == dref;
    if (inside) {                                                // uniform: digit 1 from the conditional histogram
      __shared__ unsigned long long npf2[NQ];
      __shared__ int nrem2[NQ];
      for (int q = wv; q < NQ; q += 4) locate_bin(hist[1], nrem[q], npf[q], shift - 8, lane, &npf2[q], &nrem2[q]);
      __syncthreads();
      if (tid < NQ) {
        npf[tid] = npf2[tid];
        nrem[tid] = nrem2[tid];
      }
      __syncthreads();
      done = 2u;
    }
  }
  if (tid < NQ) {
    st.prefix[tid] = npf[tid];
    st.rem[tid] = nrem[tid];
    int r = tid;
    for (int q = tid - 1; q >= 0; --q)
      if (npf[q] == npf[tid]) r = q;
    st.rep[tid] = r;
  }
  if (tid == 0) {
    st.digits_done = done;
    atomicExch(&st.arrive, 0u);
  }
}

// 1024 threads x 32 keys: up to 32768 survivors stay in registers.  With 8 ranks a sensor has 262144 keys and a 16-bit
// prefix bin keeps ~3 % of them per quantile: ~24 k survivors, which the 256-thread / 4096-key form re-read from
// memory in each of its five passes (176 us of a 230 us select; 18 us now: tools/probe_select_sharded.py).
constexpr int FIN_NT = 1024;
// Finisher: digits first_pass..7 of every rank in ONE launch, one block per sensor, no global
// hand-offs.  Reads the compacted buffer (FROM_B) — normally a handful of keys, <= 2048 stay in
// registers (<= 4096), more (e.g. thousands of identical values) are re-read per digit — or, for inputs of a
// single slice, the flat input itself.
template <bool FROM_B>
__global__ __launch_bounds__(FIN_NT) void select_finish_kernel(const KeyLayout kl, SelState* __restrict__ state,
                                                            int first_pass, const SelectArgs sa,
                                                            double* __restrict__ med_iqr) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long prefix[NQ];
  __shared__ int rem[NQ];
  __shared__ int rep[NQ];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  SelState& st = state[s];
  const unsigned int slots = (unsigned int)kl.blocks * (unsigned int)kl.pitch;
  const unsigned int cnt = FROM_B ? st.cnt_b : slots;
  const unsigned long long* in = FROM_B ? kl.bufb + (size_t)s * slots : kl.keys + (size_t)s * kl.pitch;
  if (tid < NQ) {
    prefix[tid] = st.prefix[tid];
    rem[tid] = st.rem[tid];
  }
  constexpr int FK = 32;                      // keys per thread the finisher keeps in registers
  const bool resident = cnt <= FK * FIN_NT;
  unsigned long long key[FK];
  if (resident) {
#pragma unroll
    for (int u = 0; u < FK; ++u) {
      const unsigned int i = tid + u * FIN_NT;
      key[u] = i < cnt ? in[i] : FILLER;
    }
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
    for (int i = tid; i < NQ * 256; i += FIN_NT) (&hist[0][0])[i] = 0u;
    __syncthreads();
    unsigned long long pf[NQ];
    bool active[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pf[q] = prefix[q];
      active[q] = rep[q] == q;
    }
    // returns false when the key matches no rank's prefix any more (it never will again)
    auto tally = [&](unsigned long long k) -> bool {
      if (k == FILLER) return false;
      const unsigned int digit = (unsigned int)(k >> shift) & 255u;
      bool any = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (!active[q]) continue;
        if (pass == 0 || ((k ^ pf[q]) >> (shift + 8)) == 0ull) {
          atomicAdd(&hist[q][digit], 1u);
          any = true;
        }
      }
      return any;
    };
    if (resident) {
      // a key that matched nothing is dropped from the register set: the later passes skip it with one compare
      // (24 k survivors x 6 ranks x 5 passes of 64-bit shifts were 60 of the finisher's 67 us)
#pragma unroll
      for (int u = 0; u < FK; ++u)
        if (!tally(key[u])) key[u] = FILLER;
    } else {
      for (unsigned int i = tid; i < cnt; i += FIN_NT) tally(in[i]);
    }
    __syncthreads();
    for (int q = wv; q < NQ; q += FIN_NT / 64) locate_bin(hist[rep[q]], rem[q], pf[q], shift, lane, &prefix[q], &rem[q]);
    __syncthreads();
  }
  if (tid == 0) write_result(prefix, sa, s, med_iqr);
}

// Whole select in ONE workgroup per sensor, for sensors whose slots fit the register file of a
// 1024-thread workgroup (32 keys per thread = 32768 slots: the single-GPU series of the bench).  No global
// histograms, no tickets, no inter-block hand-offs, one launch instead of five:
//   digits 0-1 over the keys in registers (read from HBM once; 32-bit math on the high word);
//   keys matching some rank's 16-bit prefix (a few percent) are compacted into LDS, digit 2 runs on those;
//   keys matching a 24-bit prefix (a handful) are compacted again and every rank is finished by
//   COUNTING: a thread per candidate counts the smaller / equal candidates of its bucket.
// Inputs that defeat the compaction (e.g. a constant sensor: every key matches) fall back to digit
// passes over LDS or registers, still inside this launch.
constexpr int ONE_NT = 1024;
constexpr int ONE_FK = 32;
constexpr int ONE_CAP = 8192;   // 16-bit-prefix survivors kept in LDS (64 KB)
constexpr int ONE_CAP2 = 1024;  // 24-bit-prefix survivors finished by counting

template <bool FLAT>
__global__ __launch_bounds__(ONE_NT) void select_onewg_kernel(const KeyLayout kl, const SelectArgs sa,
                                                              double* __restrict__ med_iqr) {
  __shared__ unsigned int hist[NQ][256];
  __shared__ unsigned long long prefix[NQ];
  __shared__ unsigned long long stage[ONE_CAP];
  __shared__ unsigned long long stage2[ONE_CAP2];
  __shared__ unsigned short sidx[ONE_CAP];   // slot numbers of the 16-bit-prefix survivors
  __shared__ int rem[NQ];
  __shared__ int rep[NQ];
  __shared__ unsigned int n_stage, n_stage2, dref;
  __shared__ unsigned int cnt0[2];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned int slots = (unsigned int)kl.blocks * (unsigned int)kl.pitch;
#ifdef GDN_SELECT_TIMING
  const long long tick_start = wall_clock64();
#endif
  // Only the HIGH words stay in registers (digits 0-2 and both compactions look at nothing else): 32 VGPRs
  // instead of 64 — with the full keys the kernel spilled 47 VGPRs + 172 SGPRs at its 128-register budget
  // (4 waves per SIMD) and every pass paid scratch traffic.  The few keys that survive the 16-bit compaction
  // are re-read in full (L2-resident: the sensor's row was just streamed).
  constexpr bool flat = FLAT;                              // blocks == 1 (single GPU): the sensor's row is contiguous
  // flat rows go through a buffer descriptor: per-thread offset in ONE register, the slot offset u*8 KB a
  // scalar (64-bit pointers per slot cost 64 VGPRs and spilled); reads past the row return 0
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned long long*>(kl.keys + (size_t)s * kl.pitch), 0, flat ? kl.pitch * 8 : 0, 0x00020000);
  auto key_full = [&](int u) -> unsigned long long {
    if constexpr (flat) {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, tid * 8, u * (ONE_NT * 8), 0);
      return ((unsigned long long)v.y << 32) | v.x;
    } else {
      const unsigned int i = tid + u * ONE_NT;
      const unsigned int blk = i / (unsigned int)kl.pitch, off = i - blk * (unsigned int)kl.pitch;
      return kl.keys[((size_t)blk * kl.n + s) * kl.pitch + off];
    }
  };
  auto key_slot = [&](unsigned int i) -> unsigned long long {     // slot number -> full key
    if constexpr (flat) {
      return kl.keys[(size_t)s * kl.pitch + i];
    } else {
      const unsigned int blk = i / (unsigned int)kl.pitch, off = i - blk * (unsigned int)kl.pitch;
      return kl.keys[((size_t)blk * kl.n + s) * kl.pitch + off];
    }
  };
  unsigned int khi[ONE_FK];
#pragma unroll
  for (int u = 0; u < ONE_FK; ++u) {
    unsigned int hi;
    if constexpr (flat) hi = __builtin_amdgcn_raw_buffer_load_b32(rsrc, tid * 8 + 4, u * (ONE_NT * 8), 0);
    else hi = tid + u * ONE_NT < slots ? (unsigned int)(key_full(u) >> 32) : 0u;
    khi[u] = tid + u * ONE_NT < slots ? hi : 0xffffffffu;   // filler: never a real key
  }
  if (tid < NQ) {
    prefix[tid] = 0ull;
    rem[tid] = sa.rank[tid];
  }
  if (tid == 0) {
    n_stage = 0u;
    n_stage2 = 0u;
  }
  __syncthreads();
#ifdef GDN_SELECT_TIMING
  long long tick[12];
  int nt = 0;
  tick[nt++] = wall_clock64();
#endif
  bool in_lds = false;       // digits >= 2 read the compacted survivors
  bool done = false;
  unsigned int n_keep = 0u;
  for (int pass = 0; pass < 8 && !done; ++pass) {
    const int shift = 56 - 8 * pass;
    if (tid < NQ) {
      int r = tid;
      for (int q = tid - 1; q >= 0; --q)
        if (prefix[q] == prefix[tid]) r = q;
      rep[tid] = r;
    }
    for (int i = tid; i < NQ * 256; i += ONE_NT) (&hist[0][0])[i] = 0u;
    if (pass == 0 && tid == 0) {
      cnt0[0] = cnt0[1] = 0u;
      dref = khi[0] >> 24;                 // top byte of tick 0's key: the bin of (nearly) every key
    }
    __syncthreads();
    if (pass == 0) {
      // Digit 0 without a histogram: errors of one sensor almost always share the top byte (sign + 7
      // exponent bits), so COUNT the keys below / inside the reference bin (compares and adds in registers,
      // one LDS atomic pair per wave); if every rank falls inside it the digit is known.  Otherwise (ranks in
      // several bins) the generic histogram below runs.
      const unsigned int dr = dref;
      unsigned int less = 0u, same = 0u;
#pragma unroll
      for (int u = 0; u < ONE_FK; ++u) {
        const unsigned int hi = khi[u];
        const unsigned int digit = hi >> 24;             // the filler's digit is 255: never less, never same
        less += digit < dr ? 1u : 0u;
        same += digit == dr ? 1u : 0u;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        less += __shfl_xor(less, d);
        same += __shfl_xor(same, d);
      }
      if (lane == 0) {
        atomicAdd(&cnt0[0], less);
        atomicAdd(&cnt0[1], same);
      }
      __syncthreads();
      const unsigned int tl = cnt0[0], ts = cnt0[1];
      bool inside = dr != 255u;
#pragma unroll
      for (int q = 0; q < NQ; ++q) inside = inside && (unsigned int)sa.rank[q] >= tl && (unsigned int)sa.rank[q] < tl + ts;
      if (inside) {                                        // uniform
        if (tid < NQ) {
          prefix[tid] = (unsigned long long)dr << 56;
          rem[tid] = sa.rank[tid] - (int)tl;
        }
        __syncthreads();
#ifdef GDN_SELECT_TIMING
        tick[nt++] = wall_clock64();
#endif
        continue;
      }
    }
    // One REAL loop over the ranks (not unrolled: with the 32 key slots unrolled inside six copies of every
    // branch the kernel was 34 k instructions and spilled ~110 SGPRs); ranks that share their prefix with an
    // earlier one are skipped (uniform).
    if (!in_lds && pass < 3) {
      // digits 0-2 live in the HIGH word of the key: 32-bit shifts and compares
      const unsigned int sh = 24u - 8u * (unsigned int)pass;
#pragma nounroll
      for (int q = 0; q < NQ; ++q) {
        if (rep[q] != q) continue;
        const unsigned int pfh = (unsigned int)(prefix[q] >> 32);
        unsigned int* hq = hist[q];
#pragma unroll
        for (int u = 0; u < ONE_FK; ++u) {
          const unsigned int hi = khi[u];
          const bool real = (int)hi >= 0;               // keys are non-negative doubles; the filler is all ones
          const unsigned int digit = (hi >> sh) & 255u;
          if (pass == 0) {                              // (only when the counting path above gave up)
            const unsigned int d0 = __builtin_amdgcn_readfirstlane(digit);
            if (__all(real && digit == d0)) {           // a whole wave in one bin: one add
              if (lane == 0) atomicAdd(&hq[d0], 64u);
            } else if (real) {
              atomicAdd(&hq[digit], 1u);
            }
          } else if (real && ((hi ^ pfh) >> (sh + 8u)) == 0u) {
            atomicAdd(&hq[digit], 1u);
          }
        }
      }
    } else if (!in_lds) {
#pragma nounroll
      for (int q = 0; q < NQ; ++q) {
        if (rep[q] != q) continue;
        const unsigned long long pfq = prefix[q];
        unsigned int* hq = hist[q];
#pragma unroll 4
        for (int u = 0; u < ONE_FK; ++u) {      // compaction defeated (rare): full keys re-read per digit
          const bool real = (int)khi[u] >= 0;
          const unsigned long long k = real ? key_full(u) : FILLER;
          const unsigned int digit = (unsigned int)(k >> shift) & 255u;
          if (real && ((k ^ pfq) >> (shift + 8)) == 0ull) atomicAdd(&hq[digit], 1u);
        }
      }
    } else {
#pragma nounroll
      for (int q = 0; q < NQ; ++q) {
        if (rep[q] != q) continue;
        const unsigned long long pfq = prefix[q];
        unsigned int* hq = hist[q];
        for (unsigned int i = tid; i < n_keep; i += ONE_NT) {
          const unsigned long long k = stage[i];
          const unsigned int digit = (unsigned int)(k >> shift) & 255u;
          if (((k ^ pfq) >> (shift + 8)) == 0ull) atomicAdd(&hq[digit], 1u);
        }
      }
    }
    __syncthreads();
    for (int q = wv; q < NQ; q += ONE_NT / 64)
      locate_bin(hist[rep[q]], rem[q], prefix[q], shift, lane, &prefix[q], &rem[q]);
    __syncthreads();
    if (pass == 1) {
      // compact the keys that still match some rank's 16-bit prefix; stay in registers if they do not fit
      unsigned int p16[NQ];                            // distinct 16-bit prefixes as scalars; 0x10000 = unused
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        p16[q] = __builtin_amdgcn_readfirstlane((unsigned int)(prefix[q] >> 48));
#pragma unroll
        for (int r = 0; r < q; ++r)
          if (p16[r] == p16[q]) p16[q] = 0x10000u;
      }
      unsigned int keep = 0u;                          // bit u: my key u survives
#pragma unroll
      for (int u = 0; u < ONE_FK; ++u) {
        const unsigned int h16 = khi[u] >> 16;   // the filler (0xffff) never equals a prefix
        bool any = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) any |= h16 == p16[q];
        keep |= any ? (1u << u) : 0u;
      }
      // one LDS atomic per WAVE: lanes scan their survivor counts, the last lane reserves the range
      const int mine = __popc(keep);
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      unsigned int base = 0u;
      if (lane == 63) base = atomicAdd(&n_stage, (unsigned int)incl);
      base = __shfl(base, 63);
      unsigned int pos = base + (unsigned int)(incl - mine);
      // survivors first as SLOT NUMBERS, then re-read in full by all threads at once (a load inside the
      // per-slot branch would pay one L2 round trip per survivor, serially)
      unsigned int t1 = tid;
      asm volatile("" : "+v"(t1));     // keeps the 32 slot numbers from being precomputed outside the pass loop
#pragma unroll
      for (int u = 0; u < ONE_FK; ++u) {
        if ((keep >> u) & 1u) {
          if (pos < (unsigned int)ONE_CAP) sidx[pos] = (unsigned short)(t1 + u * ONE_NT);   // slots <= 32768
          ++pos;
        }
      }
      __syncthreads();
      n_keep = n_stage;
      in_lds = n_keep <= (unsigned int)ONE_CAP;
      if (in_lds) {
        for (unsigned int i = tid; i < n_keep; i += ONE_NT) stage[i] = key_slot(sidx[i]);
        __syncthreads();
      }
    }
    if (pass == 2 && in_lds) {
      // second compaction (24-bit prefixes), then finish every rank by counting inside its bucket
      for (unsigned int i = tid; i < n_keep; i += ONE_NT) {
        const unsigned long long k = stage[i];
        bool any = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) any |= ((k ^ prefix[q]) >> 40) == 0ull;
        if (any) {
          const unsigned int at = atomicAdd(&n_stage2, 1u);
          if (at < (unsigned int)ONE_CAP2) stage2[at] = k;
        }
      }
      __syncthreads();
      const unsigned int n2 = n_stage2;
      if (n2 <= (unsigned int)ONE_CAP2) {
        if ((unsigned int)tid < n2) {
          const unsigned long long k = stage2[tid];
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const unsigned long long pq = prefix[q];
            if (((k ^ pq) >> 40) != 0ull) continue;
            int less = 0, equal = 0;
            for (unsigned int j = 0; j < n2; ++j) {
              const unsigned long long o = stage2[j];
              if (((o ^ pq) >> 40) != 0ull) continue;
              less += o < k ? 1 : 0;
              equal += o == k ? 1 : 0;
            }
            const int want = rem[q];                    // 0-based position inside the bucket
            if (less <= want && want < less + equal) prefix[q] = k;   // equal keys write the same value
          }
        }
        __syncthreads();
        done = true;
      }
    }
#ifdef GDN_SELECT_TIMING
    tick[nt++] = wall_clock64();
#endif
  }
#ifdef GDN_SELECT_TIMING
  if (tid == 0 && s == 1)
    printf("onewg ticks(10ns): load %lld p0 %lld p1+compact %lld p2+finish %lld (passes run %d) survivors %u / %u\n",
           tick[0] - tick_start, tick[1] - tick[0], tick[2] - tick[1], tick[3] - tick[2], nt - 1, n_keep, n_stage2);
#endif
  if (tid == 0) write_result(prefix, sa, s, med_iqr);
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
      // (x - median) / (|iqr| + eps), evaluate.py:60-62: the divisor is one number per sensor, so the wave divides
      // ONCE per sensor and run and multiplies per tick (a float64 division is ~35 instructions; eleven per sensor
      // and run were a third of this kernel).  x * (1/d) is within one ulp of x / d: 2e-16 relative, four orders
      // below the 1e-12 bar of the scoring parity tests.
      const double meda = med_iqr[2 * ca], dena = 1.0 / (fabs(med_iqr[2 * ca + 1]) + 1e-2);
      const double medb = med_iqr[2 * cb], denb = 1.0 / (fabs(med_iqr[2 * cb + 1]) + 1e-2);
      // ALL 4 x (3 + RUN) values of the run are fetched before the first is used (clamped row indices: the loads are
      // unconditional, rows that do not exist are masked afterwards) — fetched inside the tick loop they were a
      // chain of dependent round trips and the kernel ran at 2.2 TB/s
      const int g0 = first_tick + t0;
      float pa[RUN + 3], ga[RUN + 3], pb[RUN + 3], gb[RUN + 3];
#pragma unroll
      for (int u = 0; u < RUN + 3; ++u) {
        const int tt = t0 - 3 + u;                 // < 0: halo row 3 + tt (rows before this shard's first tick)
        const bool halo = tt < 0;
        const int row = halo ? (first_tick > 0 ? 3 + tt : 0) : min(tt, t - 1);
        const float* pp = (halo && first_tick > 0 ? halo_pred : pred) + (size_t)row * n;
        const float* gg = (halo && first_tick > 0 ? halo_gt : gt) + (size_t)row * n;
        pa[u] = pp[ca]; ga[u] = gg[ca]; pb[u] = pp[cb]; gb[u] = gg[cb];
      }
      auto norm = [&](float pv, float gv, double med, double inv_den) -> double {
        return (fabs((double)pv - (double)gv) - med) * inv_den;
      };
      // the 3 predecessors of t0 that exist in the series (a missing one only feeds ticks whose
      // series index is < 3, which are forced to 0)
      double a3 = g0 >= 3 ? norm(pa[0], ga[0], meda, dena) : 0.0;
      double a2 = g0 >= 2 ? norm(pa[1], ga[1], meda, dena) : 0.0;
      double a1 = g0 >= 1 ? norm(pa[2], ga[2], meda, dena) : 0.0;
      double b3 = g0 >= 3 ? norm(pb[0], gb[0], medb, denb) : 0.0;
      double b2 = g0 >= 2 ? norm(pb[1], gb[1], medb, denb) : 0.0;
      double b1 = g0 >= 1 ? norm(pb[2], gb[2], medb, denb) : 0.0;
      double mt[RUN];                             // this lane's max over its sensors, per tick of the run
#pragma unroll
      for (int u = 0; u < RUN; ++u) {
        const int tick = t0 + u;
        mt[u] = -INFINITY;
        if (tick < t1) {                           // (wave uniform)
          const double a0 = norm(pa[3 + u], ga[3 + u], meda, dena);
          const double b0 = norm(pb[3 + u], gb[3 + u], medb, denb);
          double sma = 0.0, smb = 0.0;
          if (first_tick + tick >= 3) {               // numpy sums the 4 values left to right
            sma = (((a3 + a2) + a1) + a0) / 4.0;
            smb = (((b3 + b2) + b1) + b0) / 4.0;
          }
          if (scores) {
            if (la) scores[(size_t)sa_ * t + tick] = sma;
            if (lb) scores[(size_t)sb_ * t + tick] = smb;
          }
          mt[u] = fmax(la ? sma : -INFINITY, lb ? smb : -INFINITY);
          a3 = a2; a2 = a1; a1 = a0;
          b3 = b2; b2 = b1; b1 = b0;
        }
      }
      // max over the 64 lanes for all 8 ticks together: a butterfly that also transposes — each step a lane keeps
      // half of its ticks and hands the other half to its partner (8 + 4 + 2 exchanges), then three plain steps on
      // the one tick left: 20 64-bit exchanges instead of 8 x 6 = 48 (max is exact in any order)
      static_assert(RUN == 8, "the transposing reduction is written for 8 ticks");
      double k4[4], k2[2], k1;
      {
        const bool up = (lane & 32) != 0;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          const double mine = up ? mt[4 + i2] : mt[i2], send = up ? mt[i2] : mt[4 + i2];
          k4[i2] = fmax(mine, __shfl_xor(send, 32));
        }
      }
      {
        const bool up = (lane & 16) != 0;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const double mine = up ? k4[2 + i2] : k4[i2], send = up ? k4[i2] : k4[2 + i2];
          k2[i2] = fmax(mine, __shfl_xor(send, 16));
        }
      }
      {
        const bool up = (lane & 8) != 0;
        const double mine = up ? k2[1] : k2[0], send = up ? k2[0] : k2[1];
        k1 = fmax(mine, __shfl_xor(send, 8));
      }
      k1 = fmax(k1, __shfl_xor(k1, 4));
      k1 = fmax(k1, __shfl_xor(k1, 2));
      k1 = fmax(k1, __shfl_xor(k1, 1));
      if ((lane & 7) == 0) {                       // lanes 0, 8, .., 56 hold ticks 0, 1, .., 7 of the run
        const int u = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
        best[wv][u] = s0 == 0 ? k1 : fmax(best[wv][u], k1);
      }
    }
    // (LDS writes above and reads below are by the same wave, in program order)
    if (lane < t1 - t0) anomaly[t0 + lane] = best[wv][lane];
  }
}

}  // namespace

namespace {

SelectArgs make_select_args(long long t) {
  SelectArgs sa;
  // np.median: middle value, or the mean of the two middle values when t is even
  sa.median_pair = (t % 2 == 0);
  sa.rank[0] = (int)(sa.median_pair ? t / 2 - 1 : (t - 1) / 2);
  sa.rank[1] = (int)(sa.median_pair ? t / 2 : (t - 1) / 2);
  // np.percentile(method='linear'): virtual index n*q + (alpha + q*(1-alpha-beta)) - 1, alpha=beta=1
  const double qs[2] = {25.0 / 100.0, 75.0 / 100.0};
  for (int h = 0; h < 2; ++h) {
    const double vi = (double)t * qs[h] + (1.0 + qs[h] * (1.0 - 1.0 - 1.0)) - 1.0;
    const double lo = floor(vi);
    long long ilo = (long long)lo, ihi = ilo + 1;
    if (ilo < 0) ilo = 0;
    if (ihi > t - 1) ihi = t - 1;
    if (ilo > t - 1) ilo = t - 1;
    sa.rank[2 + 2 * h] = (int)ilo;
    sa.rank[3 + 2 * h] = (int)ihi;
    sa.gamma[h] = vi - lo;
  }
  return sa;
}

// diagnostic knob: GDN_SELECT_MULTI=1 forces the multi-launch path at every size (tests, comparisons)
bool force_multi_block() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("GDN_SELECT_MULTI");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

int run_select(const double* keys, int blocks, int n, int pitch, long long total, double* workspace,
               double* med_iqr, hipStream_t st) {
  const long long slots = (long long)blocks * pitch;
  KeyLayout kl;
  kl.keys = reinterpret_cast<const unsigned long long*>(keys);
  kl.bufb = reinterpret_cast<unsigned long long*>(workspace);
  kl.blocks = blocks; kl.n = n; kl.pitch = pitch;
  SelState* state = reinterpret_cast<SelState*>(kl.bufb + (size_t)slots * n);
  const SelectArgs sa = make_select_args(total);
  const int slices = (int)((slots + SLICE - 1) / SLICE);
  if (slices > 1 && slots <= (long long)ONE_NT * ONE_FK && !force_multi_block()) {
    // the whole sensor fits one workgroup's registers: one launch, no global state
    if (kl.blocks == 1) hipLaunchKernelGGL(select_onewg_kernel<true>, dim3(n), dim3(ONE_NT), 0, st, kl, sa, med_iqr);
    else hipLaunchKernelGGL(select_onewg_kernel<false>, dim3(n), dim3(ONE_NT), 0, st, kl, sa, med_iqr);
    return gdn_launch_status();
  }
  hipLaunchKernelGGL(select_init_kernel, dim3(n), dim3(256), 0, st, state, sa);
  if (slices == 1) {   // tiny input: everything in the finisher, straight from the input
    hipLaunchKernelGGL(select_finish_kernel<false>, dim3(n), dim3(FIN_NT), 0, st, kl, state, 0, sa, med_iqr);
    return gdn_launch_status();
  }
  hipLaunchKernelGGL((select_pass_kernel<false, false>), dim3(slices, n), dim3(256), 0, st, kl, state, 0);
  hipLaunchKernelGGL((select_pass_kernel<false, false>), dim3(slices, n), dim3(256), 0, st, kl, state, 1);
  hipLaunchKernelGGL((select_pass_kernel<false, true>), dim3(slices, n), dim3(256), 0, st, kl, state, 2);
  // digits 3-7 on the survivors (normally a few percent of the keys) in one launch per sensor
  hipLaunchKernelGGL(select_finish_kernel<true>, dim3(n), dim3(FIN_NT), 0, st, kl, state, 3, sa, med_iqr);
  return gdn_launch_status();
}

}  // namespace

// select workspace: compacted-key buffer [n][blocks*pitch] + one SelState per sensor
extern "C" long long gdn_score_select_workspace_bytes(int blocks, int n, int pitch) {
  if (blocks <= 0 || n <= 0 || pitch <= 0) return 0;
  return (long long)blocks * pitch * n * 8 + (long long)n * (long long)sizeof(SelState) + 256;
}

// quantiles workspace: the key buffer [n][t] + the select workspace
extern "C" long long gdn_score_workspace_bytes(int t, int n) {
  if (t <= 0 || n <= 0) return 0;
  return (long long)t * n * 8 + gdn_score_select_workspace_bytes(1, n, t);
}

extern "C" int gdn_score_keys(const float* pred, const float* gt, int t, int n, int pitch, double* keys,
                              void* stream) {
  if (!pred || !gt || !keys || t <= 0 || n <= 0 || pitch < t) return GDN_ERR_ARG;
  hipLaunchKernelGGL(score_keys_kernel, dim3((pitch + 63) / 64, (n + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, pred, gt, t, n, pitch, keys);
  return gdn_launch_status();
}

extern "C" int gdn_score_select(const double* keys, int blocks, int n, int pitch, long long total,
                                double* workspace, double* med_iqr, void* stream) {
  if (!keys || !workspace || !med_iqr || blocks <= 0 || n <= 0 || pitch <= 0 || total <= 0) return GDN_ERR_ARG;
  if (total > (long long)blocks * pitch || (long long)blocks * pitch > 0x7fffffffll) return GDN_ERR_ARG;
  if (blocks > 1 && pitch % SLICE != 0) return GDN_ERR_UNSUPPORTED;   // slices must not straddle blocks
  return run_select(keys, blocks, n, pitch, total, workspace, med_iqr, (hipStream_t)stream);
}

extern "C" int gdn_score_quantiles(const float* pred, const float* gt, int t, int n, double* workspace,
                                   double* med_iqr, void* stream) {
  if (!pred || !gt || !workspace || !med_iqr || t <= 0 || n <= 0) return GDN_ERR_ARG;
  double* keys = workspace;
  const int rc = gdn_score_keys(pred, gt, t, n, t, keys, stream);
  if (rc != GDN_OK) return rc;
  return run_select(keys, 1, n, t, t, workspace + (size_t)t * n, med_iqr, (hipStream_t)stream);
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
