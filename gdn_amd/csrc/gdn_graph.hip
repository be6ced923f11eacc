// Sensor-graph construction and per-forward constants (run once per weight update;
// microseconds of work, so the kernels favour exactness and determinism over speed).
#include "gdn_common.hpp"

#include <mutex>

// ------------------------------------------------------------------ cached device properties (thread safe)
namespace {
struct OccKey {
  const void* fn;
  int dev, threads, lds, blocks;
};
std::mutex g_prop_mutex;
OccKey g_occ[256];
int g_occ_count = 0;
int g_cus[32] = {0};
}  // namespace

int gdn_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
  std::lock_guard<std::mutex> lock(g_prop_mutex);
  if (g_cus[dev] == 0) {
    hipDeviceProp_t prop;
    g_cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0
                     ? prop.multiProcessorCount : 256;
  }
  return g_cus[dev];
}

int gdn_blocks_per_cu(const void* fn, int threads, int lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(g_prop_mutex);
  bool seen = false;                      // the dynamic-LDS attribute is per (kernel, device): set it once
  for (int i = 0; i < g_occ_count; ++i) {
    if (g_occ[i].fn != fn || g_occ[i].dev != dev) continue;
    seen = true;
    if (g_occ[i].threads == threads && g_occ[i].lds == lds) return g_occ[i].blocks;
  }
  if (!seen && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    (void)hipGetLastError();
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb <= 0) {
    (void)hipGetLastError();
    nb = 1;
  }
  if (g_occ_count < 256) g_occ[g_occ_count++] = {fn, dev, threads, lds, nb};
  return nb;
}

extern "C" int gdn_abi_version(void) { return GDN_ABI_VERSION; }
extern "C" int gdn_nbr_pitch(int k) { return ((k + 1) + 15) & ~15; }

namespace {

// "a ranks before b" in the descending order torch.topk uses: larger first, NaN largest;
// ties go to the lower index (torch leaves tie order unspecified; this is our rule).
__device__ __forceinline__ bool ranks_before(float a, int ia, float b, int ib) {
  const bool an = a != a, bn = b != b;
  if (an || bn) {
    if (an && bn) return ia < ib;
    return an;
  }
  if (a > b) return true;
  if (a < b) return false;
  return ia < ib;
}

// Neighbour list of target i from the ranks of all sensors (rank < k = member of top-k):
// entries != i keep their rank order, then i itself (models/graph_layer.py:61-63), padding = the
// sentinel index n (the kernels give it weight 0).
__device__ __forceinline__ void emit_list(const int* __restrict__ rank_of, int i, int n, int k,
                                          int pitch, uint16_t* __restrict__ nbr_row,
                                          int32_t* __restrict__ deg_i) {
  const int self_rank = rank_of[i];
  const int nonself = k - (self_rank < k ? 1 : 0);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int r = rank_of[j];
    if (r < k && j != i) nbr_row[r - (self_rank < r ? 1 : 0)] = (uint16_t)j;
  }
  if (threadIdx.x == 0) nbr_row[nonself] = (uint16_t)i;
  for (int p = nonself + 1 + threadIdx.x; p < pitch; p += blockDim.x) nbr_row[p] = (uint16_t)n;  // sentinel
  if (threadIdx.x == 0) *deg_i = nonself + 1;
}

// One workgroup per target sensor i: reference models/GDN.py:148-159 (cosine row, top-k),
// then the list form of :161-163.
__device__ __forceinline__ void graph_row(
    const float* __restrict__ emb, int n, int d, int k, int pitch, int64_t* __restrict__ topk_idx,
    uint16_t* __restrict__ nbr, int32_t* __restrict__ deg, float* __restrict__ cos_out, float* smem_graph) {
  float* cosrow = smem_graph;                            // [n]
  int* rank_of = reinterpret_cast<int*>(smem_graph + n);  // [n]
  const int i = blockIdx.x;
  const float* ei = emb + (size_t)i * d;

  // rows are read as float4 (d is a multiple of 4: 16 B aligned rows) — with scalar loads a thread issued d
  // dependent L2 round trips and the kernel took 17 us of a 230 us training step; the k-ordered fmaf chains
  // (and therefore every cosine bit) are unchanged
  float ni2 = 0.f;
  for (int t = 0; t < d; ++t) ni2 = fmaf(ei[t], ei[t], ni2);
  const float ni = sqrtf(ni2);
  const float4* ei4 = reinterpret_cast<const float4*>(ei);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float4* ej4 = reinterpret_cast<const float4*>(emb + (size_t)j * d);
    float dot = 0.f, nj2 = 0.f;
    for (int t = 0; t < d / 4; ++t) {
      const float4 a4 = ei4[t], b4 = ej4[t];
      dot = fmaf(a4.x, b4.x, dot); nj2 = fmaf(b4.x, b4.x, nj2);
      dot = fmaf(a4.y, b4.y, dot); nj2 = fmaf(b4.y, b4.y, nj2);
      dot = fmaf(a4.z, b4.z, dot); nj2 = fmaf(b4.z, b4.z, nj2);
      dot = fmaf(a4.w, b4.w, dot); nj2 = fmaf(b4.w, b4.w, nj2);
    }
    const float c = dot / (ni * sqrtf(nj2));  // GDN.py:152 has no epsilon: zero rows give NaN
    cosrow[j] = c;
    if (cos_out) cos_out[(size_t)i * n + j] = c;
  }
  __syncthreads();
  // rank by counting: deterministic, n^2 compares per row, no sorting network.  With fewer sensors than threads
  // the count of one candidate is split over P = blockDim / n threads (integer counts: order does not matter).
  const int P = (int)blockDim.x / n;
  if (P <= 1) {
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const float cj = cosrow[j];
      int rank = 0;
      for (int l = 0; l < n; ++l) rank += ranks_before(cosrow[l], l, cj, j) ? 1 : 0;
      rank_of[j] = rank;
      if (rank < k) topk_idx[(size_t)i * k + rank] = j;
    }
  } else {
    for (int j = threadIdx.x; j < n; j += blockDim.x) rank_of[j] = 0;
    __syncthreads();
    if ((int)threadIdx.x < P * n) {
      const int j = threadIdx.x % n, part = threadIdx.x / n;
      const int len = (n + P - 1) / P, l0 = part * len, l1 = min(n, l0 + len);
      const float cj = cosrow[j];
      int cnt = 0;
      for (int l = l0; l < l1; ++l) cnt += ranks_before(cosrow[l], l, cj, j) ? 1 : 0;
      atomicAdd(&rank_of[j], cnt);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      const int rank = rank_of[j];
      if (rank < k) topk_idx[(size_t)i * k + rank] = j;
    }
  }
  __syncthreads();
  emit_list(rank_of, i, n, k, pitch, nbr + (size_t)i * pitch, deg + i);
}

__global__ __launch_bounds__(256) void gdn_graph_kernel(
    const float* __restrict__ emb, int n, int d, int k, int pitch, int64_t* __restrict__ topk_idx,
    uint16_t* __restrict__ nbr, int32_t* __restrict__ deg, float* __restrict__ cos_out) {
  extern __shared__ float smem_graph[];
  graph_row(emb, n, d, k, pitch, topk_idx, nbr, deg, cos_out, smem_graph);
}

// Same list build from a given [n,k] top-k table.  The table is caller data: entries outside
// [0, n) and repeated entries are dropped (first occurrence wins), so every slot below deg is
// written and deg counts what was kept — the aggregation kernels never see an unwritten index.
__global__ __launch_bounds__(256) void gdn_graph_from_topk_kernel(
    const int64_t* __restrict__ topk_idx, int n, int k, int pitch, uint16_t* __restrict__ nbr,
    int32_t* __restrict__ deg) {
  extern __shared__ float smem_graph[];
  int* rank_of = reinterpret_cast<int*>(smem_graph);  // [n]: lowest rank naming j, or k
  const int i = blockIdx.x;
  for (int j = threadIdx.x; j < n; j += blockDim.x) rank_of[j] = k;  // "not in top-k"
  __syncthreads();
  for (int r = threadIdx.x; r < k; r += blockDim.x) {
    const int64_t j = topk_idx[(size_t)i * k + r];
    if (j >= 0 && j < n) atomicMin(&rank_of[j], r);
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // once per graph, k <= 1023: a serial compaction keeps rank order
    uint16_t* row = nbr + (size_t)i * pitch;
    int p = 0;
    for (int r = 0; r < k; ++r) {
      const int64_t j = topk_idx[(size_t)i * k + r];
      if (j >= 0 && j < n && j != i && rank_of[j] == r) row[p++] = (uint16_t)j;
    }
    row[p++] = (uint16_t)i;   // models/graph_layer.py:61-63: self loop appended last
    deg[i] = p;
    for (; p < pitch; ++p) row[p] = (uint16_t)n;
  }
}

// a_i = lin^T att_i, a_j = lin^T att_j (zero padded to 64), c_i[s] = v_s.att_em_i, c_j[s].
__device__ __forceinline__ void node_terms_block(
    const float* __restrict__ lin_w, const float* __restrict__ att_i, const float* __restrict__ att_j,
    const float* __restrict__ att_em_i, const float* __restrict__ att_em_j,
    const float* __restrict__ emb, int n, int d, int w, float* __restrict__ out, int block) {
  const int t = block * blockDim.x + threadIdx.x;
  if (t < 2 * GDN_A_PITCH) {
    const int col = t % GDN_A_PITCH;
    const float* att = t < GDN_A_PITCH ? att_i : att_j;
    float acc = 0.f;
    if (col < w)
      for (int r = 0; r < d; ++r) acc = fmaf(lin_w[(size_t)r * w + col], att[r], acc);
    out[t] = acc;
  } else if (t < 2 * GDN_A_PITCH + 2 * n) {
    const int u = t - 2 * GDN_A_PITCH;
    const int s = u % n;
    const float* att = u < n ? att_em_i : att_em_j;
    float acc = 0.f;
    for (int r = 0; r < d; ++r) acc = fmaf(emb[(size_t)s * d + r], att[r], acc);
    out[t] = acc;
  }
}

__global__ __launch_bounds__(256) void gdn_node_terms_kernel(
    const float* __restrict__ lin_w, const float* __restrict__ att_i, const float* __restrict__ att_j,
    const float* __restrict__ att_em_i, const float* __restrict__ att_em_j,
    const float* __restrict__ emb, int n, int d, int w, float* __restrict__ out) {
  node_terms_block(lin_w, att_i, att_j, att_em_i, att_em_j, emb, n, d, w, out, blockIdx.x);
}

// One launch for the two parameter-only prologues of a training step (every dependent launch inside a
// captured step costs ~5 us whatever it does): blocks [0, n) build the graph rows, the rest the folded terms.
__global__ __launch_bounds__(256) void gdn_graph_terms_kernel(
    const float* __restrict__ emb, int n, int d, int k, int pitch, int64_t* __restrict__ topk_idx,
    uint16_t* __restrict__ nbr, int32_t* __restrict__ deg, const float* __restrict__ lin_w,
    const float* __restrict__ att_i, const float* __restrict__ att_j, const float* __restrict__ att_em_i,
    const float* __restrict__ att_em_j, int w, float* __restrict__ terms) {
  extern __shared__ float smem_graph[];
  if ((int)blockIdx.x < n) graph_row(emb, n, d, k, pitch, topk_idx, nbr, deg, nullptr, smem_graph);
  else node_terms_block(lin_w, att_i, att_j, att_em_i, att_em_j, emb, n, d, w, terms, (int)blockIdx.x - n);
}

__global__ void gdn_bn_fold_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                                   const float* __restrict__ mean, const float* __restrict__ var,
                                   float eps, int c, float* __restrict__ affine) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < c) {
    const float scale = weight[t] / sqrtf(var[t] + eps);
    affine[t] = scale;
    affine[c + t] = bias[t] - mean[t] * scale;
  }
}

}  // namespace

extern "C" int gdn_topk_graph(const float* emb, int n, int d, int k, int64_t* topk_idx, uint16_t* nbr,
                              int32_t* deg, float* cos_out, void* stream) {
  if (!emb || !topk_idx || !nbr || !deg || n <= 0 || d <= 0 || k <= 0) return GDN_ERR_ARG;
  if (k > n || n > 4096 || k + 1 > 1024 || (d & 3)) return GDN_ERR_UNSUPPORTED;
  const int pitch = gdn_nbr_pitch(k);
  hipLaunchKernelGGL(gdn_graph_kernel, dim3(n), dim3(256), 2 * n * sizeof(float), (hipStream_t)stream,
                     emb, n, d, k, pitch, topk_idx, nbr, deg, cos_out);
  return gdn_launch_status();
}

extern "C" int gdn_graph_from_topk(const int64_t* topk_idx, int n, int k, uint16_t* nbr, int32_t* deg,
                                   void* stream) {
  if (!topk_idx || !nbr || !deg || n <= 0 || k <= 0) return GDN_ERR_ARG;
  if (k > n || n > 4096 || k + 1 > 1024) return GDN_ERR_UNSUPPORTED;
  const int pitch = gdn_nbr_pitch(k);
  hipLaunchKernelGGL(gdn_graph_from_topk_kernel, dim3(n), dim3(256), n * sizeof(int),
                     (hipStream_t)stream, topk_idx, n, k, pitch, nbr, deg);
  return gdn_launch_status();
}

extern "C" int gdn_node_terms(const float* lin_w, const float* att_i, const float* att_j,
                              const float* att_em_i, const float* att_em_j, const float* emb, int n,
                              int d, int w, float* node_terms, void* stream) {
  if (!lin_w || !att_i || !att_j || !att_em_i || !att_em_j || !emb || !node_terms || n <= 0 ||
      d <= 0 || w <= 0)
    return GDN_ERR_ARG;
  if (w > GDN_MAX_W) return GDN_ERR_UNSUPPORTED;
  const int total = 2 * GDN_A_PITCH + 2 * n;
  hipLaunchKernelGGL(gdn_node_terms_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, lin_w, att_i, att_j, att_em_i, att_em_j, emb, n, d, w,
                     node_terms);
  return gdn_launch_status();
}

extern "C" int gdn_topk_graph_terms(const float* emb, int n, int d, int k, int64_t* topk_idx, uint16_t* nbr,
                                    int32_t* deg, const float* lin_w, const float* att_i, const float* att_j,
                                    const float* att_em_i, const float* att_em_j, int w, float* node_terms,
                                    void* stream) {
  if (!emb || !topk_idx || !nbr || !deg || !lin_w || !att_i || !att_j || !att_em_i || !att_em_j || !node_terms ||
      n <= 0 || d <= 0 || k <= 0 || w <= 0)
    return GDN_ERR_ARG;
  if (k > n || n > 4096 || k + 1 > 1024 || (d & 3) || w > GDN_MAX_W) return GDN_ERR_UNSUPPORTED;
  const int total = 2 * GDN_A_PITCH + 2 * n;
  hipLaunchKernelGGL(gdn_graph_terms_kernel, dim3(n + (total + 255) / 256), dim3(256), 2 * n * sizeof(float),
                     (hipStream_t)stream, emb, n, d, k, gdn_nbr_pitch(k), topk_idx, nbr, deg, lin_w, att_i, att_j,
                     att_em_i, att_em_j, w, node_terms);
  return gdn_launch_status();
}

extern "C" int gdn_bn_fold(const float* weight, const float* bias, const float* running_mean,
                           const float* running_var, float eps, int c, float* affine, void* stream) {
  if (!weight || !bias || !running_mean || !running_var || !affine || c <= 0) return GDN_ERR_ARG;
  hipLaunchKernelGGL(gdn_bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     weight, bias, running_mean, running_var, eps, c, affine);
  return gdn_launch_status();
}
