// ORACLE (test infrastructure): extern "C" surface used ONLY by tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg.  The product never links this.
#include "ofield.hpp"
#include "ocircle.hpp"
#include "offt.hpp"
#include "oblake2s.hpp"
#include "omerkle.hpp"
#include "ochannel.hpp"
using namespace orc;

extern "C" {

void orc_m31_mul(const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; i++) o[i] = (M31(a[i]) * M31(b[i])).v;
}
void orc_m31_inv(const uint32_t* a, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; i++) o[i] = M31(a[i]).inverse().v;
}
void orc_qm31_mul(const uint32_t* a, const uint32_t* b, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; i++) {
    QM31 x = QM31::from_u32(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
    QM31 y = QM31::from_u32(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
    (x * y).to_u32(o + 4 * i);
  }
}
void orc_qm31_inv(const uint32_t* a, uint32_t* o, size_t n) {
  for (size_t i = 0; i < n; i++)
    QM31::from_u32(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]).inverse().to_u32(o + 4 * i);
}
// domain point i (natural order) of CanonicCoset(log).circle_domain()
void orc_domain_point(uint32_t log, uint64_t i, uint32_t* xy) {
  PointM p = CanonicCoset(log).circle_domain().at(i);
  xy[0] = p.x.v; xy[1] = p.y.v;
}
void orc_interpolate(uint32_t* vals, uint32_t log) {
  std::vector<M31> v((size_t)1 << log);
  for (size_t i = 0; i < v.size(); i++) v[i] = M31(vals[i]);
  v = interpolate(v);
  for (size_t i = 0; i < v.size(); i++) vals[i] = v[i].v;
}
void orc_evaluate(const uint32_t* coeffs, uint32_t log_in, uint32_t* out, uint32_t log_out) {
  std::vector<M31> c((size_t)1 << log_in);
  for (size_t i = 0; i < c.size(); i++) c[i] = M31(coeffs[i]);
  std::vector<M31> v = evaluate(c, log_out);
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i].v;
}
void orc_eval_at_point(const uint32_t* coeffs, uint32_t log, const uint32_t* pt, uint32_t* out) {
  std::vector<M31> c((size_t)1 << log);
  for (size_t i = 0; i < c.size(); i++) c[i] = M31(coeffs[i]);
  PointQ p{QM31::from_u32(pt[0], pt[1], pt[2], pt[3]), QM31::from_u32(pt[4], pt[5], pt[6], pt[7])};
  eval_at_point(c, p).to_u32(out);
}
void orc_blake2s256(const uint8_t* data, size_t len, uint8_t* out) {
  Hash32 h = blake2s256(data, len);
  memcpy(out, h.data(), 32);
}
void orc_b2s_compress(uint32_t* h, const uint32_t* m, uint32_t t0, uint32_t t1, uint32_t f0, uint32_t f1) {
  b2s_compress(h, m, t0, t1, f0, f1);
}
// columns: concatenated; col_logs[c] gives each column's log size. Writes root (32 B).
// If layers_out != NULL it receives all layers concatenated from layer max_log down to 0.
void orc_merkle_commit(const uint32_t* data, const uint32_t* col_logs, size_t n_cols, uint8_t* root,
                       uint8_t* layers_out) {
  std::vector<Column> cols(n_cols);
  size_t off = 0;
  for (size_t c = 0; c < n_cols; c++) {
    size_t n = (size_t)1 << col_logs[c];
    cols[c].resize(n);
    for (size_t i = 0; i < n; i++) cols[c][i] = M31::raw(data[off + i]);
    off += n;
  }
  std::vector<const Column*> ptrs;
  for (auto& c : cols) ptrs.push_back(&c);
  MerkleProver mp = MerkleProver::commit(ptrs);
  memcpy(root, mp.root().data(), 32);
  if (layers_out) {
    size_t o = 0;
    for (int l = (int)mp.layers.size() - 1; l >= 0; l--) {
      memcpy(layers_out + o, mp.layers[l].data(), mp.layers[l].size() * 32);
      o += mp.layers[l].size() * 32;
    }
  }
}
// Channel scripted test: ops encoded as a tiny bytecode; returns final digest + drawn felts.
// op 0: mix_u64(v)  op 1: mix_u32s(words...)  op 2: draw_felt -> out  op 3: mix_root(32B as 8 words)
void orc_channel_script(const uint64_t* ops, size_t n_ops, uint8_t* digest_out, uint32_t* felts_out) {
  Channel ch;
  size_t i = 0, fo = 0;
  while (i < n_ops) {
    uint64_t op = ops[i++];
    if (op == 0) ch.mix_u64(ops[i++]);
    else if (op == 1) {
      size_t n = ops[i++];
      std::vector<uint32_t> w(n);
      for (size_t k = 0; k < n; k++) w[k] = (uint32_t)ops[i++];
      ch.mix_u32s(w);
    } else if (op == 2) {
      ch.draw_felt().to_u32(felts_out + fo);
      fo += 4;
    } else if (op == 3) {
      Hash32 r;
      for (int k = 0; k < 8; k++) { uint32_t w = (uint32_t)ops[i++]; memcpy(r.data() + 4 * k, &w, 4); }
      ch.mix_root(r);
    }
  }
  memcpy(digest_out, ch.digest.data(), 32);
}
uint64_t orc_grind(const uint8_t* digest, uint32_t bits) {
  Channel ch;
  memcpy(ch.digest.data(), digest, 32);
  return grind(ch, bits);
}
}
