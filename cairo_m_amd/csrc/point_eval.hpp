// Host-side point evaluator of the AIR (stwo FrameworkComponent::evaluate_constraint_quotients_at_point):
// used by the prover's OODS sanity check.  Lives in its own translation unit because instantiating all 34
// component descriptions over QM31 dominates compile time.
#pragma once
#include "field.hpp"
#include "air/components.hpp"

namespace cm {

struct HostRelations {
  QM31 z[air::N_RELATIONS], alpha_pow[air::N_RELATIONS][air::MAX_REL_SIZE];
};
// QM31 from 4 coordinate values that are themselves QM31 (sampled values of the 4 coordinate polynomials)
QM31 combine_ef(const QM31* c4);
// sum_k coeff[k] * C_k at the sampled mask values of component `cid`
QM31 point_eval(int cid, const QM31* tr, const QM31* it, const QM31* pp, const HostRelations& rel, const QM31* coeff,
                int n_base, QM31 shift);

}  // namespace cm
