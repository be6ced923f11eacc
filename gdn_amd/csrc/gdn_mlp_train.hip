// Train-mode OutLayer MLP (reference models/GDN.py:27-56 with out_layer_num > 1, under model.train()):
//   A_0 = act [rows, d_in]                                   (the head's activation after dropout)
//   Y_l = A_l W_l^T + b_l;  A_{l+1} = relu(BatchNorm_train(Y_l))      l = 0 .. layers-2
//   out = A_{layers-1} w_o + b_o
// and its backward (the autograd graph behind train.py:72), hand-written for gfx950.
//
// Arithmetic: fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) — no
// 16-bit splits, so the gradients carry no scaling assumptions; batch statistics and every column
// reduction in fp64, summed in a fixed order (per-workgroup partials, then one pass over them): the
// results are bitwise reproducible, no atomics anywhere.
//
// Memory: only the PRE-BatchNorm outputs Y_l are kept for the backward ([rows, hidden] fp32 each).  The
// activations A_{l+1} are never materialised: wherever one is an operand (the next layer's GEMM, the
// weight-gradient GEMM, the final dot product) it is rebuilt from Y_l while the tile is staged into LDS
// (one FMA + max per element against per-column constants), which saves one [rows, hidden] write and
// two reads per layer.
//
// Kernels:
//   mlp_gemm_kernel      C[M,N] = A(M,K) B(K,N) on 64x64 tiles, k-steps of 16 through LDS (pitch 17:
//                        conflict-free operand reads), global loads of step s+1 issued before the MFMAs of
//                        step s.  Operands are addressed by (row stride, col stride) so the same kernel is
//                        the forward (A W^T), the data gradient (dY W) and — split over the reduction,
//                        partial products reduced in a fixed order — the weight gradient (dY^T A).
//   mlp_col_kernel       streaming column passes over [rows, hidden]: the final dot product, the BatchNorm
//                        backward sums, the BatchNorm backward itself.
//   mlp_finish_kernel    folds the per-workgroup partials into constants / gradients.
#include "gdn_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 64, TN = 64, TK = 16, LP = TK + 1;   // tile sizes, LDS pitch

struct GemmArgs {
  const float* A; long long sam, sak;   // A(m, k) = A[m*sam + k*sak]
  const float* B; long long sbk, sbn;   // B(k, n) = B[k*sbk + n*sbn]
  float* C;                             // C[z][m*N + n]  (z = reduction slice; one slice unless split)
  int M, N, K;                          // K = whole reduction length
  int kslice;                           // reduction elements per slice (multiple of TK)
  const float* tr_sc; const float* tr_sh;   // relu(v*sc[c] + sh[c]) applied to an operand while staging it
  const float* bias;                    // [N] added in the epilogue, or null
  double* colstats;                     // [tiles_m][2][N]: per row-tile column sums of C and C^2, or null
  int vec;                              // 1: every extent along a contiguous direction is a multiple of 4 (float4
                                        // staging loads); 0: element loads with their own bounds (odd hidden widths)
};

// A_KC: A is contiguous along k (else along m).  B_KC: B is contiguous along k (else along n).
// TR: 0 none, 1 = transform A by its k index (forward: the previous layer's BatchNorm+ReLU),
//     2 = transform B by its n index (weight gradient: the layer input rebuilt from the stored Y).
template <bool A_KC, bool B_KC, int TR>
__global__ __launch_bounds__(256) void mlp_gemm_kernel(const GemmArgs g) {
  __shared__ float As[TM * LP];
  __shared__ float Bs[TN * LP];
  __shared__ double red[2 * 2 * TN];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wv >> 1, wn = wv & 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int kbeg = blockIdx.z * g.kslice;
  const int kend = min(g.K, kbeg + g.kslice);

  // staging coordinates: 4 consecutive elements along the contiguous direction per thread
  const int a_r = A_KC ? (tid >> 2) : ((tid & 15) * 4);   // m (first of 4 when !A_KC)
  const int a_k = A_KC ? ((tid & 3) * 4) : (tid >> 4);    // k (first of 4 when A_KC)
  const int b_r = B_KC ? (tid >> 2) : ((tid & 15) * 4);   // n
  const int b_k = B_KC ? ((tid & 3) * 4) : (tid >> 4);

  static_assert(TR != 1 || A_KC, "the A transform is indexed by k along the staged quad");
  static_assert(TR != 2 || !B_KC, "the B transform is indexed by n along the staged quad");
  auto load_a = [&](int k0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int m = m0 + a_r, k = k0 + a_k;
    if (!g.vec) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mm = A_KC ? m : m + q, kk = A_KC ? k + q : k;
        if (mm < g.M && kk < kend) {
          e[q] = g.A[(long long)mm * g.sam + (long long)kk * g.sak];
          if constexpr (TR == 1) e[q] = fmaxf(fmaf(e[q], g.tr_sc[kk], g.tr_sh[kk]), 0.f);
        }
      }
      return make_float4(e[0], e[1], e[2], e[3]);
    }
    if (m < g.M && k < kend) {
      if constexpr (A_KC) v = *reinterpret_cast<const float4*>(g.A + (long long)m * g.sam + k);
      else v = *reinterpret_cast<const float4*>(g.A + (long long)k * g.sak + m);
      if constexpr (TR == 1) {
        const float4 sc = *reinterpret_cast<const float4*>(g.tr_sc + k);
        const float4 sh = *reinterpret_cast<const float4*>(g.tr_sh + k);
        v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
      }
    }
    return v;
  };
  auto load_b = [&](int k0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n = n0 + b_r, k = k0 + b_k;
    if (!g.vec) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = B_KC ? n : n + q, kk = B_KC ? k + q : k;
        if (nn < g.N && kk < kend) {
          e[q] = g.B[(long long)kk * g.sbk + (long long)nn * g.sbn];
          if constexpr (TR == 2) e[q] = fmaxf(fmaf(e[q], g.tr_sc[nn], g.tr_sh[nn]), 0.f);
        }
      }
      return make_float4(e[0], e[1], e[2], e[3]);
    }
    if (n < g.N && k < kend) {
      if constexpr (B_KC) v = *reinterpret_cast<const float4*>(g.B + (long long)n * g.sbn + k);
      else v = *reinterpret_cast<const float4*>(g.B + (long long)k * g.sbk + n);
      if constexpr (TR == 2) {
        const float4 sc = *reinterpret_cast<const float4*>(g.tr_sc + n);
        const float4 sh = *reinterpret_cast<const float4*>(g.tr_sh + n);
        v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
      }
    }
    return v;
  };
  auto store_a = [&](const float4& v) {
    if constexpr (A_KC) {
      float* p = As + a_r * LP + a_k;
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    } else {
      float* p = As + a_r * LP + a_k;
      p[0] = v.x; p[LP] = v.y; p[2 * LP] = v.z; p[3 * LP] = v.w;
    }
  };
  auto store_b = [&](const float4& v) {
    if constexpr (B_KC) {
      float* p = Bs + b_r * LP + b_k;
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    } else {
      float* p = Bs + b_r * LP + b_k;
      p[0] = v.x; p[LP] = v.y; p[2 * LP] = v.z; p[3 * LP] = v.w;
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra = load_a(kbeg), rb = load_b(kbeg);
  const float* ap = As + (wm * 32 + l32) * LP + h;
  const float* bp = Bs + (wn * 32 + l32) * LP + h;
  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    store_a(ra);
    store_b(rb);
    __syncthreads();
    if (k0 + TK < kend) {   // next step's global loads land under this step's MFMAs
      ra = load_a(k0 + TK);
      rb = load_b(k0 + TK);
    }
#pragma unroll
    for (int j = 0; j < TK / 2; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * j], bp[2 * j], acc, 0, 0, 0);
    __syncthreads();
  }

  // epilogue: register r of lane l = C[m0 + wm*32 + (r&3) + 8(r>>2) + 4h][n0 + wn*32 + l32]
  const int n = n0 + wn * 32 + l32;
  const float bias = (g.bias && n < g.N) ? g.bias[n] : 0.f;
  float* C = g.C + (size_t)blockIdx.z * g.M * g.N;
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m < g.M && n < g.N) {
      const float v = acc[r] + bias;
      C[(size_t)m * g.N + n] = v;
      s += (double)v;
      q = fma((double)v, (double)v, q);
    }
  }
  if (g.colstats) {
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (h == 0) {
      red[(wm * 2 + 0) * TN + wn * 32 + l32] = s;
      red[(wm * 2 + 1) * TN + wn * 32 + l32] = q;
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int which = tid / TN, c = tid % TN;
      if (n0 + c < g.N)
        g.colstats[((size_t)blockIdx.y * 2 + which) * g.N + n0 + c] = red[which * TN + c] + red[(2 + which) * TN + c];
    }
  }
}

// ---------------------------------------------------------------- column passes over [rows, H]
// Thread = 4 consecutive columns; LPR = H/4 rounded up to a power of two (<= 64, so a row never
// straddles a wave; lanes past column H idle); a workgroup takes 256/LPR rows per round and strides
// over the rows.
enum { CP_OUT = 0, CP_BSTAT = 1, CP_BAPPLY = 2 };

struct ColArgs {
  const float* Y;        // [rows, H] pre-BatchNorm
  const float* consts;   // [4][H]: sc = gamma*rstd, sh = beta - mean*sc, mean, rstd
  const float* w_o;      // [H]   (CP_OUT; CP_BSTAT/BAPPLY with the rank-1 gradient of the last hidden layer)
  const float* b_o;      // [1]
  const float* d_out;    // [rows]  rank-1 gradient source: dA[m][c] = d_out[m] * w_o[c]
  const float* dA;       // [rows, H] gradient of the activation (deeper layers), or null
  const float* bmeans;   // [2][H]: mean of dYhat, mean of dYhat*yhat (CP_BAPPLY)
  float* out;            // [rows]      (CP_OUT)
  float* dY;             // [rows, H]   (CP_BAPPLY)
  double* partial;       // [grid][NQ][H] (+ [grid] scalars after it for CP_BSTAT)
  int rows, H;
};

// four consecutive columns c0 .. c0+3 of a row of H: one float4 when H is a multiple of 4 (every quad whole and
// aligned), element accesses with their own bound otherwise (columns past H read 0 / are not written)
__device__ __forceinline__ void ldf4(const float* row, int c0, int H, float (&v)[4]) {
  if ((H & 3) == 0) {
    const float4 t = *reinterpret_cast<const float4*>(row + c0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = c0 + q < H ? row[c0 + q] : 0.f;
  }
}
__device__ __forceinline__ void stf4(float* row, int c0, int H, const float (&v)[4]) {
  if ((H & 3) == 0) {
    *reinterpret_cast<float4*>(row + c0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (c0 + q < H) row[c0 + q] = v[q];
  }
}

__host__ __device__ inline int col_lanes(int H) {
  int l = 1;
  while (l * 4 < H) l <<= 1;
  return l;
}

template <int MODE>
__global__ __launch_bounds__(256) void mlp_col_kernel(const ColArgs a) {
  __shared__ double red[256 * 4];
  const int tid = threadIdx.x;
  const int lpr = col_lanes(a.H), slots = 256 / lpr;
  const int lr = tid % lpr, slot = tid / lpr, c0 = lr * 4;
  const bool live = c0 < a.H;
  float sc[4] = {}, sh[4] = {}, mu[4] = {}, is[4] = {}, wo[4] = {}, ma[4] = {}, mb[4] = {};
  if (live) {
    ldf4(a.consts, c0, a.H, sc);
    ldf4(a.consts + a.H, c0, a.H, sh);
    ldf4(a.consts + 2 * a.H, c0, a.H, mu);
    ldf4(a.consts + 3 * a.H, c0, a.H, is);
    if (a.w_o) ldf4(a.w_o, c0, a.H, wo);
    if (MODE == CP_BAPPLY) {
      ldf4(a.bmeans, c0, a.H, ma);
      ldf4(a.bmeans + a.H, c0, a.H, mb);
    }
  }
  const float bias_o = (MODE == CP_OUT) ? a.b_o[0] : 0.f;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  double s_go = 0.0;
  __shared__ float rowpart[4];                  // H > 256: a row spans two waves, their partial dot products meet here
  // (uniform trip count: the CP_OUT row reduction of wide layers has a barrier inside the loop)
  for (int mrow = blockIdx.x * slots; mrow < a.rows; mrow += gridDim.x * slots) {
    const int m = mrow + slot;
    const bool row_ok = m < a.rows;
    float y[4] = {0.f, 0.f, 0.f, 0.f}, act[4];
    if (live && row_ok) ldf4(a.Y + (size_t)m * a.H, c0, a.H, y);
#pragma unroll
    for (int v = 0; v < 4; ++v) act[v] = fmaxf(fmaf(y[v], sc[v], sh[v]), 0.f);
    if constexpr (MODE == CP_OUT) {
      float part = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) part = fmaf(act[v], wo[v], part);
      for (int st = 1; st < (lpr < 64 ? lpr : 64); st <<= 1) part += __shfl_xor(part, st);
      if (lpr > 64) {                             // 128 lanes per row: waves 2 slot and 2 slot + 1
        if ((tid & 63) == 0) rowpart[tid >> 6] = part;
        __syncthreads();
        part = rowpart[2 * slot] + rowpart[2 * slot + 1];
        __syncthreads();
      }
      if (lr == 0 && row_ok) a.out[m] = part + bias_o;
    } else {
      float gsrc[4];
      float go = 0.f;
      if (a.dA) {
        gsrc[0] = gsrc[1] = gsrc[2] = gsrc[3] = 0.f;
        if (live && row_ok) ldf4(a.dA + (size_t)m * a.H, c0, a.H, gsrc);
      } else {
        go = row_ok ? a.d_out[m] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) gsrc[v] = go * wo[v];
      }
      float dyh[4], yh[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        dyh[v] = act[v] > 0.f ? gsrc[v] : 0.f;          // ReLU'
        yh[v] = (y[v] - mu[v]) * is[v];
      }
      if constexpr (MODE == CP_BSTAT) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          s0[v] += (double)dyh[v];
          s1[v] += (double)(dyh[v] * yh[v]);
          s2[v] += (double)(go * act[v]);                // d w_o (rank-1 source only)
        }
        if (lr == 0) s_go += (double)go;
      } else {
        float o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          o[v] = sc[v] * (dyh[v] - ma[v] - yh[v] * mb[v]);   // BatchNorm backward, sc = gamma * rstd
          if (row_ok) s0[v] += (double)o[v];                 // d bias of the Linear (analytically 0)
        }
        if (live && row_ok) stf4(a.dY + (size_t)m * a.H, c0, a.H, o);
      }
    }
  }
  if constexpr (MODE != CP_OUT) {
    constexpr int NQ = MODE == CP_BSTAT ? 3 : 1;
    double* dst = a.partial + (size_t)blockIdx.x * NQ * a.H;
    const double* src[3] = {s0, s1, s2};
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn) {
      __syncthreads();
      if (live) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (c0 + v < a.H) red[slot * a.H + c0 + v] = src[qn][v];
      }
      __syncthreads();
      for (int c = tid; c < a.H; c += 256) {
        double t = 0.0;
        for (int sl = 0; sl < slots; ++sl) t += red[sl * a.H + c];
        dst[(size_t)qn * a.H + c] = t;
      }
    }
    if constexpr (MODE == CP_BSTAT) {
      __syncthreads();
      red[tid] = s_go;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += red[i];
        a.partial[(size_t)gridDim.x * NQ * a.H + blockIdx.x] = t;
      }
    }
  }
}

// ---------------------------------------------------------------- finish kernels (one thread per column)
// forward: partial column sums of Y -> BatchNorm constants, running statistics (torch: momentum update
// with the UNBIASED batch variance)
__global__ void mlp_finish_fwd_kernel(const double* __restrict__ partial, int parts, int H, double rows,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      float momentum, float* running_mean, float* running_var,
                                      long long* batches, float* __restrict__ consts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < H) {
    double s = 0.0, q = 0.0;
    for (int p = 0; p < parts; ++p) {
      s += partial[((size_t)p * 2) * H + c];
      q += partial[((size_t)p * 2 + 1) * H + c];
    }
    const double m = s / rows;
    double var = q / rows - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float scv = gamma[c] * is;
    consts[c] = scv;
    consts[H + c] = beta[c] - (float)m * scv;
    consts[2 * H + c] = (float)m;
    consts[3 * H + c] = is;
    if (running_mean && running_var) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * rows / (rows - 1.0));
    }
  }
  if (c == 0 && batches) *batches += 1;
}

__global__ void mlp_finish_bstat_kernel(const double* __restrict__ partial, int parts, int H, double rows,
                                        float* d_gamma, float* d_beta, float* d_w_o, float* d_b_o,
                                        float* __restrict__ bmeans) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < H) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int p = 0; p < parts; ++p) {
      s0 += partial[((size_t)p * 3) * H + c];
      s1 += partial[((size_t)p * 3 + 1) * H + c];
      s2 += partial[((size_t)p * 3 + 2) * H + c];
    }
    d_beta[c] = (float)s0;
    d_gamma[c] = (float)s1;
    if (d_w_o) d_w_o[c] = (float)s2;
    bmeans[c] = (float)(s0 / rows);
    bmeans[H + c] = (float)(s1 / rows);
  }
  if (c == 0 && d_b_o) {
    double t = 0.0;
    for (int p = 0; p < parts; ++p) t += partial[(size_t)parts * 3 * H + p];
    d_b_o[0] = (float)t;
  }
}

__global__ void mlp_finish_colsum_kernel(const double* __restrict__ partial, int parts, int H, float* dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < H) {
    double s = 0.0;
    for (int p = 0; p < parts; ++p) s += partial[(size_t)p * H + c];
    dst[c] = (float)s;
  }
}

// weight gradient: sum of the reduction slices in a fixed order
__global__ void mlp_finish_splitk_kernel(const float* __restrict__ partial, int slices, int count, float* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    double s = 0.0;
    for (int z = 0; z < slices; ++z) s += (double)partial[(size_t)z * count + i];
    dst[i] = (float)s;
  }
}

// ---------------------------------------------------------------- host side
constexpr int MLP_COL_GRID_MAX = 1024;
constexpr int MLP_SPLIT_MAX = 64;

bool mlp_train_shape_ok(int rows, int d_in, int hidden, int layers) {
  // hidden: any width up to 512 (the reference class's default inter_num); widths that are not a multiple of 4
  // stage their operands element by element.  d_in is the model's embedding width (16 / 32 / 64 / 128).
  return rows > 1 && layers >= 2 && layers <= 8 && d_in % 4 == 0 && d_in >= 4 && d_in <= 256 &&
         hidden >= 1 && hidden <= 512;
}

int col_grid(int rows, int H) {
  const int slots = 256 / col_lanes(H);
  const int want = (rows + slots * 8 - 1) / (slots * 8);
  return max(1, min(MLP_COL_GRID_MAX, min(want, 4 * gdn_cu_count())));
}

int split_count(int rows) {
  // reduction slices of the weight-gradient GEMM: >= 256 rows each, multiple of TK
  int s = (rows + 255) / 256;
  return max(1, min(MLP_SPLIT_MAX, s));
}

static inline size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

struct Layout {   // byte offsets
  size_t y_bytes, consts_bytes;        // per layer, in `saved`
  size_t ws_partial, ws_bmeans, ws_dy, ws_da, ws_split, ws_total;
};

Layout make_layout(int rows, int d_in, int hidden, int layers) {
  Layout L;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.y_bytes = up((size_t)rows * hidden * 4);
  L.consts_bytes = up((size_t)4 * hidden * 4);
  const int tiles_m = (rows + TM - 1) / TM;
  const size_t part = max_sz((size_t)tiles_m * 2 * hidden, (size_t)MLP_COL_GRID_MAX * (3 * hidden + 1)) * 8;
  size_t off = 0;
  L.ws_partial = off; off += up(part);
  L.ws_bmeans = off; off += up((size_t)2 * hidden * 4);
  L.ws_dy = off; off += up((size_t)rows * hidden * 4);
  L.ws_da = off; off += (layers > 2 ? up((size_t)rows * hidden * 4) : 0);
  L.ws_split = off; off += up((size_t)MLP_SPLIT_MAX * hidden * max_sz((size_t)d_in, (size_t)hidden) * 4);
  L.ws_total = off;
  return L;
}

// eval mode: BatchNorm constants from the RUNNING statistics: [sc | sh | mean | rstd] as the column passes expect
__global__ void mlp_eval_consts_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ rm, const float* __restrict__ rv, float eps, int H,
                                       float* __restrict__ consts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const float rstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  const float sc = gamma[c] * rstd;
  consts[c] = sc;
  consts[H + c] = beta[c] - rm[c] * sc;
  consts[2 * H + c] = rm[c];
  consts[3 * H + c] = rstd;
}

template <bool A_KC, bool B_KC, int TR>
void launch_gemm(const GemmArgs& g, int slices, hipStream_t st) {
  dim3 grid((g.N + TN - 1) / TN, (g.M + TM - 1) / TM, slices);
  hipLaunchKernelGGL((mlp_gemm_kernel<A_KC, B_KC, TR>), grid, dim3(256), 0, st, g);
}

}  // namespace

extern "C" long long gdn_mlp_train_saved_bytes(int rows, int d_in, int hidden, int layers) {
  if (!mlp_train_shape_ok(rows, d_in, hidden, layers)) return 0;
  const Layout L = make_layout(rows, d_in, hidden, layers);
  return (long long)((L.y_bytes + L.consts_bytes) * (size_t)(layers - 1));
}

extern "C" long long gdn_mlp_train_workspace_bytes(int rows, int d_in, int hidden, int layers) {
  if (!mlp_train_shape_ok(rows, d_in, hidden, layers)) return 0;
  return (long long)make_layout(rows, d_in, hidden, layers).ws_total;
}

// params[l] for hidden layer l: {W [hidden, K_l], b [hidden], gamma [hidden], beta [hidden]} (K_0 = d_in,
// K_l = hidden); running[l] = {running_mean, running_var} (either may be null); batches[l] =
// num_batches_tracked or null.  All arrays of pointers live in HOST memory, the pointers are device ones.
extern "C" int gdn_mlp_train_fwd(const float* act, const float* const* params, float* const* running,
                                 long long* const* batches, const float* eps, const float* momentum,
                                 const float* out_w, const float* out_b, int rows, int d_in, int hidden,
                                 int layers, void* saved, void* workspace, float* out, void* stream) {
  if (!act || !params || !eps || !momentum || !out_w || !out_b || !saved || !workspace || !out) return GDN_ERR_ARG;
  if (rows <= 1 || layers < 2) return GDN_ERR_ARG;
  if (!mlp_train_shape_ok(rows, d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const Layout L = make_layout(rows, d_in, hidden, layers);
  char* sv = static_cast<char*>(saved);
  char* ws = static_cast<char*>(workspace);
  double* partial = reinterpret_cast<double*>(ws + L.ws_partial);
  const int tiles_m = (rows + TM - 1) / TM;
  const float* in = act;
  const float* in_consts = nullptr;
  for (int l = 0; l + 1 < layers; ++l) {
    const int K = l == 0 ? d_in : hidden;
    float* Y = reinterpret_cast<float*>(sv + (size_t)l * (L.y_bytes + L.consts_bytes));
    float* consts = reinterpret_cast<float*>(sv + (size_t)l * (L.y_bytes + L.consts_bytes) + L.y_bytes);
    const float* const* p = params + 4 * l;
    if (!p[0] || !p[1] || !p[2] || !p[3]) return GDN_ERR_ARG;
    GemmArgs g = {};
    g.vec = (hidden & 3) == 0;
    g.A = in; g.sam = K; g.sak = 1;
    g.B = p[0]; g.sbk = 1; g.sbn = K;          // B(k, n) = W[n][k]
    g.C = Y; g.M = rows; g.N = hidden; g.K = K; g.kslice = K;
    g.tr_sc = in_consts; g.tr_sh = in_consts ? in_consts + hidden : nullptr;
    g.bias = p[1]; g.colstats = partial;
    if (in_consts) launch_gemm<true, true, 1>(g, 1, st);
    else launch_gemm<true, true, 0>(g, 1, st);
    float* rm = running ? running[2 * l] : nullptr;
    float* rv = running ? running[2 * l + 1] : nullptr;
    hipLaunchKernelGGL(mlp_finish_fwd_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, partial, tiles_m,
                       hidden, (double)rows, p[2], p[3], eps[l], momentum[l], rm, rv,
                       batches ? batches[l] : nullptr, consts);
    in = Y;
    in_consts = consts;
  }
  ColArgs c = {};
  c.Y = in; c.consts = in_consts; c.w_o = out_w; c.b_o = out_b; c.out = out; c.rows = rows; c.H = hidden;
  hipLaunchKernelGGL((mlp_col_kernel<CP_OUT>), dim3(col_grid(rows, hidden)), dim3(256), 0, st, c);
  return gdn_launch_status();
}

// Eval-mode OutLayer MLP (models/GDN.py:45-56 under model.eval()) on the kernels above, for the widths the
// one-launch register-resident chain of gdn_mlp.hip does not take (hidden > 256): per hidden layer one fp32
// matrix-core GEMM (+ bias), the BatchNorm (running statistics) + ReLU applied while the next GEMM stages its
// operand, then the Linear(hidden -> 1) column pass.  workspace: two [rows, hidden] buffers + two constant rows.
extern "C" long long gdn_mlp_eval_workspace_bytes(int rows, int d_in, int hidden, int layers) {
  if (rows < 1 || !mlp_train_shape_ok(rows > 1 ? rows : 2, d_in, hidden, layers)) return 0;
  const size_t y = ((size_t)rows * hidden * 4 + 255) & ~(size_t)255, cst = ((size_t)4 * hidden * 4 + 255) & ~(size_t)255;
  return (long long)(2 * y + 2 * cst);
}

extern "C" int gdn_mlp_eval_fwd(const float* act, const float* const* params, const float* const* running,
                                const float* eps, const float* out_w, const float* out_b, int rows, int d_in,
                                int hidden, int layers, void* workspace, float* out, void* stream) {
  if (!act || !params || !running || !eps || !out_w || !out_b || !workspace || !out || rows < 1) return GDN_ERR_ARG;
  if (!mlp_train_shape_ok(rows > 1 ? rows : 2, d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const size_t y = ((size_t)rows * hidden * 4 + 255) & ~(size_t)255, cst = ((size_t)4 * hidden * 4 + 255) & ~(size_t)255;
  char* ws = static_cast<char*>(workspace);
  const float* in = act;
  const float* in_consts = nullptr;
  for (int l = 0; l + 1 < layers; ++l) {
    const int K = l == 0 ? d_in : hidden;
    float* Y = reinterpret_cast<float*>(ws + (size_t)(l & 1) * y);
    float* consts = reinterpret_cast<float*>(ws + 2 * y + (size_t)(l & 1) * cst);
    const float* const* p = params + 4 * l;
    if (!p[0] || !p[1] || !p[2] || !p[3] || !running[2 * l] || !running[2 * l + 1]) return GDN_ERR_ARG;
    GemmArgs g = {};
    g.vec = (hidden & 3) == 0;
    g.A = in; g.sam = K; g.sak = 1;
    g.B = p[0]; g.sbk = 1; g.sbn = K;
    g.C = Y; g.M = rows; g.N = hidden; g.K = K; g.kslice = (K + TK - 1) / TK * TK;
    g.tr_sc = in_consts; g.tr_sh = in_consts ? in_consts + hidden : nullptr;
    g.bias = p[1]; g.colstats = nullptr;
    if (in_consts) launch_gemm<true, true, 1>(g, 1, st);
    else launch_gemm<true, true, 0>(g, 1, st);
    hipLaunchKernelGGL(mlp_eval_consts_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, p[2], p[3], running[2 * l],
                       running[2 * l + 1], eps[l], hidden, consts);
    in = Y;
    in_consts = consts;
  }
  ColArgs c = {};
  c.Y = in; c.consts = in_consts; c.w_o = out_w; c.b_o = out_b; c.out = out; c.rows = rows; c.H = hidden;
  hipLaunchKernelGGL((mlp_col_kernel<CP_OUT>), dim3(col_grid(rows, hidden)), dim3(256), 0, st, c);
  return gdn_launch_status();
}

// grads[l] = {dW, db, dgamma, dbeta} of hidden layer l (device pointers, host array).  d_act [rows, d_in]
// = gradient of the head's activation (input of gdn_head_train_bwd_act).
extern "C" int gdn_mlp_train_bwd(const float* d_out, const float* act, const float* const* params,
                                 const float* out_w, int rows, int d_in, int hidden, int layers,
                                 const void* saved, void* workspace, float* const* grads, float* d_out_w,
                                 float* d_out_b, float* d_act, void* stream) {
  if (!d_out || !act || !params || !out_w || !saved || !workspace || !grads || !d_out_w || !d_out_b || !d_act)
    return GDN_ERR_ARG;
  if (rows <= 1 || layers < 2) return GDN_ERR_ARG;
  if (!mlp_train_shape_ok(rows, d_in, hidden, layers)) return GDN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const Layout L = make_layout(rows, d_in, hidden, layers);
  const char* sv = static_cast<const char*>(saved);
  char* ws = static_cast<char*>(workspace);
  double* partial = reinterpret_cast<double*>(ws + L.ws_partial);
  float* bmeans = reinterpret_cast<float*>(ws + L.ws_bmeans);
  float* dY = reinterpret_cast<float*>(ws + L.ws_dy);
  float* dA = reinterpret_cast<float*>(ws + L.ws_da);
  float* split = reinterpret_cast<float*>(ws + L.ws_split);
  const int cgrid = col_grid(rows, hidden);
  const int fin_grid = (hidden + 255) / 256;
  for (int l = layers - 2; l >= 0; --l) {
    const int K = l == 0 ? d_in : hidden;
    const float* Y = reinterpret_cast<const float*>(sv + (size_t)l * (L.y_bytes + L.consts_bytes));
    const float* consts = reinterpret_cast<const float*>(sv + (size_t)l * (L.y_bytes + L.consts_bytes) + L.y_bytes);
    const float* const* p = params + 4 * l;
    float* const* gr = grads + 4 * l;
    if (!p[0] || !gr[0] || !gr[1] || !gr[2] || !gr[3]) return GDN_ERR_ARG;
    const bool last = l == layers - 2;     // gradient source: d_out (x) w_o, else dA of the layer above
    ColArgs c = {};
    c.Y = Y; c.consts = consts; c.rows = rows; c.H = hidden; c.partial = partial;
    if (last) { c.d_out = d_out; c.w_o = out_w; } else { c.dA = dA; }
    hipLaunchKernelGGL((mlp_col_kernel<CP_BSTAT>), dim3(cgrid), dim3(256), 0, st, c);
    hipLaunchKernelGGL(mlp_finish_bstat_kernel, dim3(fin_grid), dim3(256), 0, st, partial, cgrid, hidden,
                       (double)rows, gr[2], gr[3], last ? d_out_w : nullptr, last ? d_out_b : nullptr, bmeans);
    c.bmeans = bmeans; c.dY = dY;
    hipLaunchKernelGGL((mlp_col_kernel<CP_BAPPLY>), dim3(cgrid), dim3(256), 0, st, c);
    hipLaunchKernelGGL(mlp_finish_colsum_kernel, dim3(fin_grid), dim3(256), 0, st, partial, cgrid, hidden, gr[1]);
    // dW[h][k] = sum_m dY[m][h] * A_l[m][k]   (A_0 = act; A_l = relu(bn(Y_{l-1})) rebuilt while staging)
    {
      const int slices = split_count(rows);
      int kslice = (rows + slices - 1) / slices;
      kslice = (kslice + TK - 1) / TK * TK;
      const int used = (rows + kslice - 1) / kslice;
      GemmArgs g = {};
      g.vec = (hidden & 3) == 0;
      g.A = dY; g.sam = 1; g.sak = hidden;        // A(m' = h, k' = row) = dY[row][h]
      g.M = hidden; g.N = K; g.K = rows; g.kslice = kslice;
      g.C = used > 1 ? split : gr[0];
      if (l == 0) {
        g.B = act; g.sbk = K; g.sbn = 1;
        launch_gemm<false, false, 0>(g, used, st);
      } else {
        const float* pc = reinterpret_cast<const float*>(sv + (size_t)(l - 1) * (L.y_bytes + L.consts_bytes) + L.y_bytes);
        g.B = reinterpret_cast<const float*>(sv + (size_t)(l - 1) * (L.y_bytes + L.consts_bytes));
        g.sbk = K; g.sbn = 1;
        g.tr_sc = pc; g.tr_sh = pc + hidden;
        launch_gemm<false, false, 2>(g, used, st);
      }
      if (used > 1) {
        const int count = hidden * K;
        hipLaunchKernelGGL(mlp_finish_splitk_kernel, dim3((count + 255) / 256), dim3(256), 0, st, split, used,
                           count, gr[0]);
      }
    }
    // dA_l[m][k] = sum_h dY[m][h] * W[h][k]
    {
      GemmArgs g = {};
      g.vec = (hidden & 3) == 0;
      g.A = dY; g.sam = hidden; g.sak = 1;
      g.B = p[0]; g.sbk = K; g.sbn = 1;
      g.C = l == 0 ? d_act : dA; g.M = rows; g.N = K; g.K = hidden; g.kslice = hidden;
      launch_gemm<true, false, 0>(g, 1, st);
    }
  }
  return gdn_launch_status();
}
