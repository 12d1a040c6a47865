// Supervised normaliser statistics on the GPU (the row next to the hot path):
// per class bin b and input column j the weighted median and the weighted mean absolute deviation about it,
// i.e. what AffineNormalizer.fit computes with weighted_quantile + a matmul (_affine_normalizer.py:72-79,
// _weighted_quantile.py:35-63).  The reference sorts every column of every bin on the CPU (argsort + cumsum +
// interp, 7.6 s of 13 s at n = 2e5, d = 128); here the rows are gathered bin by bin into a column-major image, all
// d x nbins segments are sorted by one segmented radix sort (hipCUB / rocPRIM), and one workgroup per segment finds
// the two interpolated crossings of the cumulative weight and the deviation sum.  HBM-bound integer/compare work:
// no matrix pipe involved.
#include <hipcub/hipcub.hpp>

#include "nls_host.h"

namespace {

// Kt[j][p] = X[perm[p]][j],  Vt[j][p] = p   (p = position in bin-grouped order).  64 x 64 tiles through LDS so that both
// the row gathers (d contiguous doubles) and the column-major stores are coalesced.
__global__ void k_gather_transpose(const double* X, const int* perm, long n, int d, int jbase, int dc, double* Kt, int* Vt) {
  __shared__ double tile[64][65];
  const long p0 = (long)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;  // column inside the group [jbase, jbase + dc)
  for (int r = threadIdx.y; r < 64; r += blockDim.y) {
    const long p = p0 + r;
    const int j = j0 + threadIdx.x;
    tile[r][threadIdx.x] = (p < n && j < dc) ? X[(long)perm[p] * d + jbase + j] : 0.0;
  }
  __syncthreads();
  for (int c = threadIdx.y; c < 64; c += blockDim.y) {
    const int j = j0 + c;
    const long p = p0 + threadIdx.x;
    if (j < dc && p < n) {
      Kt[(long)j * n + p] = tile[threadIdx.x][c];
      Vt[(long)j * n + p] = (int)p;
    }
  }
}

__global__ void k_gather_weights(const double* s, const int* perm, long n, double* ws) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p < n) ws[p] = s[perm[p]];
}

__global__ void k_segment_offsets(const long* bin_off, int nbins, int d, long n, long* seg_begin, long* seg_end) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= d * nbins) return;
  const int j = idx / nbins, b = idx % nbins;
  seg_begin[idx] = (long)j * n + bin_off[b];
  seg_end[idx] = (long)j * n + bin_off[b + 1];
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// One workgroup per (column, bin) segment of sorted values a[0..m) with weights w[k] = ws[V[k]]:
//   median = (interp(1/2, p_lower, a) + interp(1/2, p_upper, a)) / 2,  p_upper = cumsum(w) / W, p_lower = p_upper - w / W
//   spread = sum w |a - median| / W
// With t = first k whose cum[k] / W > 1/2:  interp over p_lower lands in [a[t], a[t+1]], over p_upper in [a[t-1], a[t]].
__global__ void k_segment_median(const double* a_all, const int* v_all, const double* ws, const long* seg_begin, const long* seg_end,
                                 int nbins, int d, int jbase, double* centers, double* spreads) {
  __shared__ double sh[8];
  __shared__ double pre[256];
  __shared__ double res[2];
  if (threadIdx.x == 0) res[0] = __builtin_nan("");  // stays NaN when the bin has no weight (as the reference's 0/0)
  const int seg = blockIdx.x;
  const long beg = seg_begin[seg], m = seg_end[seg] - beg;
  const int j = jbase + seg / nbins, b = seg % nbins;
  if (m <= 0) {
    if (threadIdx.x == 0) centers[(long)b * d + j] = spreads[(long)b * d + j] = 0.0;
    return;
  }
  const double* a = a_all + beg;
  const int* v = v_all + beg;
  // contiguous range per thread
  const long per = (m + blockDim.x - 1) / blockDim.x;
  const long k0 = std::min<long>(m, (long)threadIdx.x * per), k1 = std::min<long>(m, k0 + per);
  double part = 0.0;
  for (long k = k0; k < k1; ++k) part += ws[v[k]];
  pre[threadIdx.x] = part;
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan of 256 partial sums (sequential: negligible)
    double run = 0.0;
    for (int i = 0; i < (int)blockDim.x; ++i) {
      const double t = pre[i];
      pre[i] = run;
      run += t;
    }
    sh[0] = run;
  }
  __syncthreads();
  const double W = sh[0];
  const double half = 0.5;
  // the thread whose range contains the crossing finds t
  double cum = pre[threadIdx.x];
  if (k1 > k0 && cum / W <= half && (cum + part) / W > half) {
    long t = k0;
    double prev = cum;
    for (long k = k0; k < k1; ++k) {
      const double c = prev + ws[v[k]];
      if (c / W > half) {
        t = k;
        cum = c;
        break;
      }
      prev = c;
    }
    const double p_hi = cum / W, p_lo = prev / W;  // p_upper[t], p_upper[t-1] == p_lower[t]
    const double at = a[t];
    double m_lo, m_hi;
    if (t + 1 < m) {  // interp over p_lower: xp[t] = p_lo <= 1/2 < xp[t+1] = p_hi
      m_lo = (a[t + 1] - at) / (p_hi - p_lo) * (half - p_lo) + at;
    } else {
      m_lo = at;
    }
    if (t > 0) {  // interp over p_upper: xp[t-1] = p_lo <= 1/2 < xp[t] = p_hi
      const double am = a[t - 1];
      m_hi = (at - am) / (p_hi - p_lo) * (half - p_lo) + am;
    } else {
      m_hi = at;
    }
    res[0] = (m_lo + m_hi) / 2;
  }
  __syncthreads();
  const double mu = res[0];
  double dev = 0.0;
  for (long k = threadIdx.x; k < m; k += blockDim.x) dev += ws[v[k]] * fabs(a[k] - mu);
  dev = block_sum(dev, sh);
  if (threadIdx.x == 0) {
    centers[(long)b * d + j] = mu;
    spreads[(long)b * d + j] = dev / W;
  }
}

}  // namespace

// Shared core: dperm (the n row indices grouped by bin, stable) and doff (nbins + 1 offsets into it) are on the device.
static int bin_stats_core(nls_ctx* ctx, const double* X, const double* s, int64_t n, int d, const int* dperm, const long* doff, int nbins,
                          double* centers, double* spreads) {
  // Column groups keep one segmented sort below 2^30 keys (hipCUB counts items in an int) and ~24 bytes/key of workspace.
  const int dg = (int)std::max<int64_t>(1, std::min<int64_t>(d, ((int64_t)1 << 30) / n));
  if ((int64_t)n > ((int64_t)1 << 30)) return fail(ctx, NLS_ERR_ARG, "nls_bin_stats: n too large");
  const long Ng = (long)n * dg;
  const double *dX = nullptr, *ds = nullptr;
  NLSCHK(resident(ctx, "in.X", X, (size_t)n * d, &dX));
  NLSCHK(resident(ctx, "in.s", s, (size_t)n, &ds));
  int *Vin = nullptr, *Vout = nullptr;
  long *sbeg = nullptr, *send = nullptr;
  double *Kin = nullptr, *Kout = nullptr, *ws = nullptr, *dcen = nullptr, *dspr = nullptr;
  NLSCHK(ws_get_t(ctx, "pre.sbeg", (size_t)dg * nbins, &sbeg));
  NLSCHK(ws_get_t(ctx, "pre.send", (size_t)dg * nbins, &send));
  NLSCHK(ws_get_t(ctx, "pre.Kin", (size_t)Ng, &Kin));
  NLSCHK(ws_get_t(ctx, "pre.Kout", (size_t)Ng, &Kout));
  NLSCHK(ws_get_t(ctx, "pre.Vin", (size_t)Ng, &Vin));
  NLSCHK(ws_get_t(ctx, "pre.Vout", (size_t)Ng, &Vout));
  NLSCHK(ws_get_t(ctx, "pre.ws", (size_t)n, &ws));
  NLSCHK(ws_get_t(ctx, "pre.cen", (size_t)d * nbins, &dcen));
  NLSCHK(ws_get_t(ctx, "pre.spr", (size_t)d * nbins, &dspr));
  hipLaunchKernelGGL(k_gather_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ds, dperm, (long)n, ws);
  HIPCHK(ctx, hipGetLastError());
  for (int j0 = 0; j0 < d; j0 += dg) {
    const int dc = std::min(dg, d - j0);
    const int nseg = dc * nbins;
    const long N = (long)n * dc;
    hipLaunchKernelGGL(k_gather_transpose, dim3((unsigned)((n + 63) / 64), (unsigned)((dc + 63) / 64)), dim3(64, 4), 0, ctx->stream, dX, dperm,
                       (long)n, d, j0, dc, Kin, Vin);
    hipLaunchKernelGGL(k_segment_offsets, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, ctx->stream, doff, nbins, dc, (long)n, sbeg,
                       send);
    HIPCHK(ctx, hipGetLastError());
    size_t temp_bytes = 0;
    HIPCHK(ctx, hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp_bytes, Kin, Kout, Vin, Vout, (int)N, nseg, sbeg, send, 0, 64,
                                                            ctx->stream));
    void* temp = nullptr;
    NLSCHK(ws_get(ctx, "pre.sort_tmp", std::max<size_t>(temp_bytes, 8), &temp));
    HIPCHK(ctx, hipcub::DeviceSegmentedRadixSort::SortPairs(temp, temp_bytes, Kin, Kout, Vin, Vout, (int)N, nseg, sbeg, send, 0, 64,
                                                            ctx->stream));
    hipLaunchKernelGGL(k_segment_median, dim3((unsigned)nseg), dim3(256), 0, ctx->stream, Kout, Vout, ws, sbeg, send, nbins, d, j0, dcen,
                       dspr);
    HIPCHK(ctx, hipGetLastError());
  }
  HIPCHK(ctx, hipMemcpyAsync(centers, dcen, sizeof(double) * d * nbins, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(spreads, dspr, sizeof(double) * d * nbins, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return NLS_OK;
}

// centers, spreads: nbins x d (host).  perm: the n row indices grouped by bin (stable), bin_off: nbins + 1 offsets into perm.
extern "C" int nls_bin_stats(nls_ctx* ctx, const double* X, const double* s, int64_t n, int d, const int32_t* perm,
                             const int64_t* bin_off, int nbins, double* centers, double* spreads) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !s || !perm || !bin_off || !centers || !spreads || n < 1 || d < 1 || nbins < 1)
    return fail(ctx, NLS_ERR_ARG, "nls_bin_stats: NULL argument or empty problem");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int* dperm = nullptr;
  long* doff = nullptr;
  NLSCHK(ws_get_t(ctx, "pre.perm", (size_t)n, &dperm));
  NLSCHK(ws_get_t(ctx, "pre.off", (size_t)nbins + 1, &doff));
  HIPCHK(ctx, hipMemcpyAsync(dperm, perm, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
  static_assert(sizeof(long) == sizeof(int64_t), "offsets are 64-bit");
  HIPCHK(ctx, hipMemcpyAsync(doff, bin_off, sizeof(long) * (nbins + 1), hipMemcpyHostToDevice, ctx->stream));
  return bin_stats_core(ctx, X, s, n, d, dperm, doff, nbins, centers, spreads);
}

namespace {
__global__ void k_iota(int* v, long n) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int)i;
}
// off[b] = first position of the sorted labels that holds a label >= b (b = 0 .. nbins): the bins' offsets into the grouped order
__global__ void k_label_offsets(const int* sorted, long n, int nbins, long* off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > nbins) return;
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (sorted[mid] < b) lo = mid + 1; else hi = mid;
  }
  off[b] = lo;
}
// rank codes of the sorted keys: flag[i] = keys[i] != keys[i - 1] (IEEE comparison: -0.0 == +0.0, as numpy.unique compares)
__global__ void k_new_value_flags(const double* keys, long n, long* flag) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i > 0 && keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void k_scatter_codes(const long* code, const int* idx, long n, long* inv) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) inv[idx[i]] = code[i];
}
}  // namespace

// The same statistics from the per-row bin labels (0 .. nbins - 1; host): the grouping - numpy's argsort(labels, kind="stable") and the
// bincount / cumsum offsets, 60-80 ms of host time at n = 1e6 - runs on the device (one stable radix sort of (label, row) pairs + nbins + 1
// binary searches): the same permutation, hence bit-identical statistics.
extern "C" int nls_bin_stats_labels(nls_ctx* ctx, const double* X, const double* s, int64_t n, int d, const int32_t* labels, int nbins,
                                    double* centers, double* spreads) {
  if (!ctx) return NLS_ERR_ARG;
  if (!X || !s || !labels || !centers || !spreads || n < 1 || d < 1 || nbins < 1)
    return fail(ctx, NLS_ERR_ARG, "nls_bin_stats_labels: NULL argument or empty problem");
  if ((int64_t)n > ((int64_t)1 << 30)) return fail(ctx, NLS_ERR_ARG, "nls_bin_stats_labels: n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int *lab_in = nullptr, *lab_out = nullptr, *idx_in = nullptr, *dperm = nullptr;
  long* doff = nullptr;
  NLSCHK(ws_get_t(ctx, "pre.lab_in", (size_t)n, &lab_in));
  NLSCHK(ws_get_t(ctx, "pre.lab_out", (size_t)n, &lab_out));
  NLSCHK(ws_get_t(ctx, "pre.idx_in", (size_t)n, &idx_in));
  NLSCHK(ws_get_t(ctx, "pre.perm", (size_t)n, &dperm));
  NLSCHK(ws_get_t(ctx, "pre.off", (size_t)nbins + 1, &doff));
  HIPCHK(ctx, hipMemcpyAsync(lab_in, labels, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, idx_in, (long)n);
  HIPCHK(ctx, hipGetLastError());
  int bits = 1;
  while (bits < 31 && (1L << bits) < (long)nbins) ++bits;
  size_t temp_bytes = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, lab_in, lab_out, idx_in, dperm, (int)n, 0, bits, ctx->stream));
  void* temp = nullptr;
  NLSCHK(ws_get(ctx, "pre.sort_tmp", std::max<size_t>(temp_bytes, 8), &temp));
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, lab_in, lab_out, idx_in, dperm, (int)n, 0, bits, ctx->stream));
  hipLaunchKernelGGL(k_label_offsets, dim3((unsigned)((nbins + 1 + 255) / 256)), dim3(256), 0, ctx->stream, lab_out, (long)n, nbins, doff);
  HIPCHK(ctx, hipGetLastError());
  return bin_stats_core(ctx, X, s, n, d, dperm, doff, nbins, centers, spreads);
}

// inverse[i] = rank of y[i] among the distinct values of y, *nunique = their number: numpy.unique(y, return_inverse=True)[1] and len(...[0]) -
// the first step of the target quantiser (sample_bins_quantized_ecdf, _quantizer.py:246-253; 48 ms of host time at n = 1e6) - by one radix
// sort of (value, row) pairs, a flag / inclusive-scan pass and a scatter.  y: finite doubles, host | device; inverse: n int64, host.
extern "C" int nls_rank_codes(nls_ctx* ctx, const double* y, int64_t n, int64_t* inverse, int64_t* nunique) {
  if (!ctx) return NLS_ERR_ARG;
  if (!y || !inverse || !nunique || n < 1) return fail(ctx, NLS_ERR_ARG, "nls_rank_codes: NULL argument or n < 1");
  if ((int64_t)n > ((int64_t)1 << 30)) return fail(ctx, NLS_ERR_ARG, "nls_rank_codes: n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double* dy = nullptr;
  NLSCHK(resident(ctx, "rank.y", y, (size_t)n, &dy));
  double* keys = nullptr;
  int *idx_in = nullptr, *idx_out = nullptr;
  long *flag = nullptr, *code = nullptr, *dinv = nullptr;
  NLSCHK(ws_get_t(ctx, "rank.keys", (size_t)n, &keys));
  NLSCHK(ws_get_t(ctx, "pre.idx_in", (size_t)n, &idx_in));
  NLSCHK(ws_get_t(ctx, "rank.idx_out", (size_t)n, &idx_out));
  NLSCHK(ws_get_t(ctx, "rank.flag", (size_t)n, &flag));
  NLSCHK(ws_get_t(ctx, "rank.code", (size_t)n, &code));
  NLSCHK(ws_get_t(ctx, "rank.inv", (size_t)n, &dinv));
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, idx_in, (long)n);
  HIPCHK(ctx, hipGetLastError());
  size_t tb1 = 0, tb2 = 0;
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, dy, keys, idx_in, idx_out, (int)n, 0, 64, ctx->stream));
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tb2, flag, code, (int)n, ctx->stream));
  void* temp = nullptr;
  NLSCHK(ws_get(ctx, "pre.sort_tmp", std::max<size_t>(std::max(tb1, tb2), 8), &temp));
  HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(temp, tb1, dy, keys, idx_in, idx_out, (int)n, 0, 64, ctx->stream));
  hipLaunchKernelGGL(k_new_value_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, keys, (long)n, flag);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(temp, tb2, flag, code, (int)n, ctx->stream));
  hipLaunchKernelGGL(k_scatter_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, code, idx_out, (long)n, dinv);
  HIPCHK(ctx, hipGetLastError());
  static_assert(sizeof(long) == sizeof(int64_t), "codes are 64-bit");
  long last = 0;
  HIPCHK(ctx, hipMemcpyAsync(inverse, dinv, sizeof(long) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(&last, code + (n - 1), sizeof(long), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  *nunique = last + 1;
  return NLS_OK;
}
