// extern "C" wrappers for the single-op backend entry points that mirror Stwo's FieldOps / FriOps /
// QuotientOps traits (declared in include/cairom_hip.h).  The whole-segment prover calls the same
// engine functions directly.
#include "../../include/cairom_hip.h"
#include "engine.hpp"
#include "fri_kernels.hpp"
#include "air_kernels.hpp"
#include <vector>
#include <string>
#include <algorithm>

using namespace cm;
extern "C" int32_t cm_set_last_error(const char* msg);

namespace {
inline hipStream_t S(cm_stream_t s) { return (hipStream_t)(uintptr_t)s; }
inline uint32_t* P32(cm_handle h) { return (uint32_t*)(uintptr_t)h; }
template <class F>
int32_t guard(F&& f) {
  try { f(); return 0; }
  catch (const CmError& e) { cm_set_last_error(e.what()); return e.code ? e.code : 1; }
  catch (const std::exception& e) { cm_set_last_error(e.what()); return 1; }
}
__global__ void k_inverse_m31(const uint32_t* in, uint32_t* out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = inv(M31(in[i])).v;
}
struct P4 { uint32_t* p[4]; };
struct CP4 { const uint32_t* p[4]; };
__global__ void k_inverse_qm31(CP4 in, P4 out, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  QM31 v = inv(QM31(M31(in.p[0][i]), M31(in.p[1][i]), M31(in.p[2][i]), M31(in.p[3][i])));
  out.p[0][i] = v.a.a.v; out.p[1][i] = v.a.b.v; out.p[2][i] = v.b.a.v; out.p[3][i] = v.b.b.v;
}
// FriOps::decompose helpers: per-block partial sums of (+f on the first half, -f on the second), then the shift
__global__ void k_decompose_sum(CP4 f, uint32_t log_n, uint32_t* partial /*[gridDim.x][4]*/) {
  __shared__ uint32_t red[4][256];
  const uint64_t n = 1ull << log_n, half = n >> 1;
  QM31 acc;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    QM31 v(M31(f.p[0][i]), M31(f.p[1][i]), M31(f.p[2][i]), M31(f.p[3][i]));
    if (i < half) acc += v; else acc = acc - v;
  }
  red[0][threadIdx.x] = acc.a.a.v; red[1][threadIdx.x] = acc.a.b.v; red[2][threadIdx.x] = acc.b.a.v; red[3][threadIdx.x] = acc.b.b.v;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int k = 0; k < 4; k++) red[k][threadIdx.x] = (M31(red[k][threadIdx.x]) + M31(red[k][threadIdx.x + s])).v;
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ void k_decompose_apply(P4 f, uint32_t log_n, const uint32_t* partial, uint32_t n_partial, uint32_t* lambda_out) {
  const uint64_t n = 1ull << log_n, half = n >> 1;
  M31 lam[4];
  for (int k = 0; k < 4; k++) {
    M31 t(0);
    for (uint32_t b = 0; b < n_partial; b++) t = t + M31(partial[b * 4 + k]);
    lam[k] = t * inv(M31::from_u32((uint32_t)n));
  }
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) for (int k = 0; k < 4; k++) lambda_out[k] = lam[k].v;
  if (i >= n) return;
  for (int k = 0; k < 4; k++) f.p[k][i] = (i < half ? M31(f.p[k][i]) - lam[k] : M31(f.p[k][i]) + lam[k]).v;
}
}  // namespace

extern "C" {

// AccumulationOps (Stwo core::air::accumulation): column += other, and the powers of the random coefficient
int32_t cm_accumulate(const cm_handle dst[4], const cm_handle src[4], uint64_t n, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(n < (1ull << 32), "cm_accumulate: column too long");
    std::vector<uint32_t*> d(4);
    std::vector<const uint32_t*> c(4);
    for (int k = 0; k < 4; k++) { d[k] = P32(dst[k]); c[k] = P32(src[k]); CM_CHECK(d[k] && c[k], "cm_accumulate: null column"); }
    DevBuf dd = upload(d, S(s)), dc = upload(c, S(s));
    add_columns(dd.as<uint32_t*>(), dc.as<const uint32_t*>(), 4, (uint32_t)n, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_generate_secure_powers(const uint32_t felt[4], uint64_t n, uint32_t* out) {
  return guard([&] {
    QM31 f = QM31::from_u32(felt), cur(M31(1));
    for (uint64_t i = 0; i < n; i++) { cur.to_u32(out + 4 * i); cur = cur * f; }
  });
}
int32_t cm_col_zero(cm_handle h, uint64_t n_u32, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(h, "cm_col_zero: null column");
    CM_HIP(hipMemsetAsync(P32(h), 0, n_u32 * 4, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}

// FriOps::decompose (Stwo core::fri; CPU backend: lambda = (sum first half - sum second half) / n on the bit-reversed
// evaluation, g = f -/+ lambda).  Not on the prove_cairo_m path (every committed column is inside the FFT space), kept
// for the Backend trait surface.
int32_t cm_fri_decompose(const cm_handle f[4], uint32_t log_n, uint32_t lambda_out[4], cm_stream_t s) {
  return guard([&] {
    CM_CHECK(log_n >= 1 && log_n <= 30, "cm_fri_decompose: bad log size");
    CP4 c; P4 m;
    for (int k = 0; k < 4; k++) { m.p[k] = P32(f[k]); c.p[k] = m.p[k]; CM_CHECK(m.p[k], "cm_fri_decompose: null column"); }
    const uint64_t n = 1ull << log_n;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 64);
    DevBuf partial((size_t)blocks * 16), lam(16);
    hipLaunchKernelGGL(k_decompose_sum, dim3(blocks), dim3(256), 0, S(s), c, log_n, partial.u32());
    hipLaunchKernelGGL(k_decompose_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(s), m, log_n, partial.u32(), blocks,
                       lam.u32());
    CM_HIP(hipGetLastError());
    CM_HIP(hipMemcpyAsync(lambda_out, lam.p, 16, hipMemcpyDeviceToHost, S(s)));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}

// FieldOps::batch_inverse: element-wise inverses (the Montgomery trick of the CPU backend is a
// latency optimisation; one Fermat chain per lane is the GPU-native form and gives the same values).
int32_t cm_batch_inverse_m31(cm_handle in, cm_handle out, uint64_t n, cm_stream_t s) {
  return guard([&] {
    if (!n) return;
    hipLaunchKernelGGL(k_inverse_m31, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(s), P32(in), P32(out), n);
    CM_HIP(hipGetLastError());
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_batch_inverse_qm31(const cm_handle in[4], const cm_handle out[4], uint64_t n, cm_stream_t s) {
  return guard([&] {
    if (!n) return;
    CP4 i4; P4 o4;
    for (int k = 0; k < 4; k++) { i4.p[k] = P32(in[k]); o4.p[k] = P32(out[k]); }
    hipLaunchKernelGGL(k_inverse_qm31, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(s), i4, o4, n);
    CM_HIP(hipGetLastError());
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_fri_fold_circle_into_line(const cm_handle dst[4], const cm_handle src[4], const uint32_t alpha[4], uint32_t log_n,
                                     cm_handle tw, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_fri_fold_circle_into_line: null twiddles");
    uint32_t* d[4]; const uint32_t* c[4];
    for (int k = 0; k < 4; k++) { d[k] = P32(dst[k]); c[k] = P32(src[k]); }
    fold_circle_into_line(d, c, log_n, *(Twiddles*)(uintptr_t)tw, QM31::from_u32(alpha), true, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_fri_fold_line(const cm_handle in[4], const uint32_t alpha[4], uint32_t log_n, cm_handle tw, const cm_handle out[4],
                         cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_fri_fold_line: null twiddles");
    uint32_t* d[4]; const uint32_t* c[4];
    for (int k = 0; k < 4; k++) { d[k] = P32(out[k]); c[k] = P32(in[k]); }
    fold_line(d, c, log_n, *(Twiddles*)(uintptr_t)tw, QM31::from_u32(alpha), S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_fri_fold_line_leaves(const cm_handle* in, const cm_handle* circle, const uint32_t* alpha, const uint32_t* alpha_circle,
                                uint32_t log_n, cm_handle tw, const cm_handle out[4], cm_handle leaf_hashes, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_fri_fold_line_leaves: null twiddles");
    CM_CHECK(log_n >= 2, "cm_fri_fold_line_leaves: log_n < 2");
    CM_CHECK(in || circle, "cm_fri_fold_line_leaves: neither a line nor a circle source");
    CM_CHECK((in == nullptr) == (alpha == nullptr) && (circle == nullptr) == (alpha_circle == nullptr),
             "cm_fri_fold_line_leaves: every source comes with its challenge");
    const Twiddles& T = *(Twiddles*)(uintptr_t)tw;
    uint32_t* d[4]; const uint32_t* c[4] = {nullptr, nullptr, nullptr, nullptr}; const uint32_t* q[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; k++) { d[k] = P32(out[k]); if (in) c[k] = P32(in[k]); if (circle) q[k] = P32(circle[k]); }
    std::vector<uint32_t> a8(8, 0);
    if (alpha) memcpy(a8.data(), alpha, 16);
    if (alpha_circle) memcpy(a8.data() + 4, alpha_circle, 16);
    DevBuf d_a = upload(a8, S(s));
    const bool fused = in ? fold_line_leaf(d, c, circle ? q : nullptr, log_n, T, S(s), d_a.u32(), circle ? d_a.u32() + 4 : nullptr, P32(leaf_hashes))
                          : fold_circle_leaf(d, q, log_n, T, S(s), d_a.u32() + 4, P32(leaf_hashes));
    if (!fused) {
      // small layers: the two launches the fused kernel replaces
      if (!in) fold_circle_into_line(d, q, log_n, T, QM31::from_u32(alpha_circle), false, S(s));
      else if (circle) fold_line_and_circle(d, c, q, log_n, T, S(s), d_a.u32(), d_a.u32() + 4);
      else fold_line(d, c, log_n, T, QM31::from_u32(alpha), S(s));
      std::vector<const uint32_t*> cols(d, d + 4);
      DevBuf d_cols = upload(cols, S(s));
      merkle_layer(log_n - 1, nullptr, d_cols.as<const uint32_t*>(), 4, P32(leaf_hashes), S(s));
    }
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
int32_t cm_accumulate_quotients(uint32_t log_size, const cm_handle* cols, uint32_t n_cols, const cm_sample_batches* b,
                                const uint32_t random_coeff[4], const cm_handle out[4], cm_handle tw, cm_stream_t s) {
  return guard([&] {
    CM_CHECK(tw, "cm_accumulate_quotients: null twiddles");
    QM31 rc = QM31::from_u32(random_coeff);
    std::vector<const uint32_t*> c(n_cols);
    for (uint32_t i = 0; i < n_cols; i++) c[i] = P32(cols[i]);
    std::vector<QuotientBatch> qb(b->n_batches);
    std::vector<uint32_t> col_index, coef_c;
    for (uint32_t k = 0; k < b->n_batches; k++) {
      CPoint<QM31> pt{QM31::from_u32(b->points + 8 * k), QM31::from_u32(b->points + 8 * k + 4)};
      qb[k].begin = (uint32_t)col_index.size();
      QM31 alpha(M31(1)), sum_a, sum_b;
      QM31 cdiff = conj_u(pt.y) - pt.y;
      for (uint32_t e = b->batch_off[k]; e < b->batch_off[k + 1]; e++) {
        QM31 v = QM31::from_u32(b->values + 4 * e);
        alpha = alpha * rc;
        QM31 a = conj_u(v) - v;
        QM31 bb = v * cdiff - a * pt.y;
        sum_a += alpha * a;
        sum_b += alpha * bb;
        col_index.push_back(b->col_index[e]);
        uint32_t w[4];
        (alpha * cdiff).to_u32(w);
        coef_c.insert(coef_c.end(), w, w + 4);
      }
      qb[k].end = (uint32_t)col_index.size();
      pt.x.to_u32(qb[k].point);
      pt.y.to_u32(qb[k].point + 4);
      sum_a.to_u32(qb[k].sum_a);
      sum_b.to_u32(qb[k].sum_b);
      qpow(rc, b->batch_off[k + 1] - b->batch_off[k]).to_u32(qb[k].batch_coeff);
    }
    std::vector<uint32_t*> o(4);
    for (int k = 0; k < 4; k++) o[k] = P32(out[k]);
    DevBuf dcols = upload(c, S(s)), dci = upload(col_index, S(s)), dcc = upload(coef_c, S(s)), dqb = upload(qb, S(s)),
           dout = upload(o, S(s));
    QuotientArgs a;
    a.tw = view(*(Twiddles*)(uintptr_t)tw); a.log_size = log_size; a.cols = dcols.as<const uint32_t*>();
    a.col_index = dci.u32(); a.coef_c = dcc.u32(); a.batches = dqb.as<QuotientBatch>(); a.n_batches = b->n_batches;
    a.out = dout.as<uint32_t*>();
    launch_quotients(a, (double)n_cols, S(s));
    CM_HIP(hipStreamSynchronize(S(s)));
  });
}
}
