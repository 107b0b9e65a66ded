#include "point_eval.hpp"

namespace cm {

// ---- host point evaluator (stwo FrameworkComponent::evaluate_constraint_quotients_at_point) ------------
struct PointEvalH : air::LogupStream<PointEvalH, QM31, QM31> {
  const QM31 *tr, *it, *pp;
  const HostRelations* rels;
  const QM31* coeff;
  int n_base;
  QM31 cumsum_shift, prev_col, acc;
  int ci = 0, ii = 0, kb = 0, kl = 0;
  QM31 next() { return tr[ci++]; }
  QM31 preproc(int id) { return pp[id]; }
  QM31 c(uint32_t v) { return QM31(M31(v)); }
  void constraint(QM31 x) { acc += coeff[kb++] * x; }
  void constraint_q(QM31 x) { acc += coeff[n_base + kl++] * x; }
  QM31 combine(int r, const QM31* v, int n) {
    QM31 a;
    for (int i = 0; i < n; i++) a += rels->alpha_pow[r][i] * v[i];
    return a - rels->z[r];
  }
  QM31 ef_from(QM31 m) { return m; }
  void on_entry(int, QM31, const QM31*, int) {}
  static QM31 combine_ef(const QM31* c4) {
    return c4[0] + c4[1] * QM31(M31(0), M31(1), M31(0), M31(0)) + c4[2] * QM31(M31(0), M31(0), M31(1), M31(0)) +
           c4[3] * QM31(M31(0), M31(0), M31(0), M31(1));
  }
  void emit_batch(bool last, QM31 num, QM31 den) {
    if (!last) {
      QM31 cur = combine_ef(it + ii);
      ii += 4;
      QM31 diff = cur - prev_col;
      prev_col = cur;
      constraint_q(diff * den - num);
    } else {
      QM31 pr[4], cu[4];
      for (int k = 0; k < 4; k++) { pr[k] = it[ii + 2 * k]; cu[k] = it[ii + 2 * k + 1]; }
      ii += 8;
      constraint_q((combine_ef(cu) - combine_ef(pr) - prev_col + cumsum_shift) * den - num);
    }
  }
};
QM31 point_eval(int cid, const QM31* tr, const QM31* it, const QM31* pp, const HostRelations& rel, const QM31* coeff,
                       int n_base, QM31 shift) {
  PointEvalH e;
  e.tr = tr; e.it = it; e.pp = pp; e.rels = &rel; e.coeff = coeff; e.n_base = n_base; e.cumsum_shift = shift;
  switch (cid) {
#define CM_X(id, T) case air::id: air::T::eval(e); break;
    AIR_ALL_COMPONENTS(CM_X)
#undef CM_X
  }
  return e.acc;
}

QM31 combine_ef(const QM31* c4) { return PointEvalH::combine_ef(c4); }

}  // namespace cm
