// FRI commit phase of the segment prover (stwo `prove`, reached from crates/prover/src/prover.rs:131): first-layer tree over the
// DEEP quotient columns, circle / line folds, one Merkle tree per inner layer, the last layer's polynomial — with the
// Fiat-Shamir steps between the layers on the device (k_chan_mix_root_draw; the host replays them afterwards and cross-checks).
// The decommitment planner of the FRI trees lives with the phase's declaration (prover_common.hpp: FriPhase::plan_decommit /
// finish_decommit, MerkleTree::plan_decommit in merkle_tree.hpp).
#include "prover_common.hpp"

namespace cm {

void FriPhase::commit_enqueue(Prover& P, const cm_pcs_config& cfg, std::vector<ColumnSet>& quotients, const std::vector<uint32_t>& q_logs,
                              const FriResume* resume) {
  hipStream_t st = P.st;
  Channel& ch = P.ch;
  KProfAloneScope kprof_alone_scope;   // (measurement only: a lone proof runs this phase one kernel at a time, kprof.hpp)
  // ---- FRI commit ----
  // The whole commit phase is enqueued without a host round trip: after each layer's Merkle tree a 1-thread
  // kernel does the transcript step (mix_root, draw the folding challenge) on a device copy of the channel,
  // and the fold kernels read the challenge from device memory.  The host replays the same steps on its own
  // channel afterwards from the recorded roots and checks that the challenges agree.
  const uint32_t last_log = cfg.log_last_layer_degree_bound + cfg.log_blowup_factor;
  uint32_t layer_log = resume ? resume->layer_log : q_logs[0] - 1;
  if (resume) { have_first = false; inner_fold0 = 1 + resume->n_inner_before; }
  const uint32_t n_inner = layer_log > last_log ? layer_log - last_log : 0;
  // challenges and roots share one buffer ({alphas | roots}: they come back in ONE copy at the end)
  d_ar.alloc((size_t)(n_inner + 1) * 48);
  n_inner_ = n_inner; last_log_ = last_log; resumed_ = resume != nullptr;
  struct Words { uint32_t* p; uint32_t* u32() const { return p; } };
  const Words d_alphas{d_ar.u32()}, d_roots{d_ar.u32() + (size_t)(n_inner + 1) * 4};
  Words d_chan{nullptr};   // the device copy of the channel {digest[8], n_sent}: travels with the tree tables (one upload)
  std::vector<uint32_t> chan_words(16, 0);
  memcpy(chan_words.data(), ch.digest.data(), 32);
  chan_words[8] = ch.n_sent;
  // every layer above the single-launch tail is allocated up front so that the column tables of all their
  // Merkle trees (and of the first-layer tree) travel in ONE host->device copy
  std::vector<std::unique_ptr<InnerLayer>> pre;
  // Separate processes could not tell the two forms apart (profiles/r05f_ab_fri_top_fuse.txt: the small layers are bound by the
  // dependent-compression chain, not by launches); alternating blocks inside one process can: -0.07 ms per proof, 13 of 16 pairs
  // (profiles/r05_ab_switches_in_process.txt).  On.
  const bool top_fuse = tune(T_FRI_TOP_FUSE) != 0;
  {
    UploadBatch ub;
    std::vector<const uint32_t*> cols;
    std::vector<uint32_t> logs;
    if (!resume) {
      for (size_t k = 0; k < quotients.size(); k++) for (int c = 0; c < 4; c++) { cols.push_back(quotients[k].ptrs[c]); logs.push_back(q_logs[k]); }
      first_tree.prepare(cols, logs);
      ub.add(first_tree.cols, &first_tree.d_cols_view);
    }
    for (uint32_t l = layer_log; l > last_log && l > fri_tail_log(); l--) {
      std::unique_ptr<InnerLayer> il(new InnerLayer());
      il->log = l;
      if (resume && resume->layer && l == layer_log) il->eval = std::move(*resume->layer);
      else il->eval.alloc(std::vector<uint32_t>(4, l), st, false);
      std::vector<const uint32_t*> lc(il->eval.ptrs.begin(), il->eval.ptrs.end());
      il->tree.prepare(lc, std::vector<uint32_t>(4, l));
      ub.add(il->tree.cols, &il->tree.d_cols_view);
      pre.push_back(std::move(il));
    }
    ub.add(chan_words, &d_chan.p);
    fri_tables = ub.flush(st);
    if (!resume) {
      // (round 5) the transcript step behind a tree — mix_root + the draw of the folding challenge — runs in the block that hashes
      // the root (MerkleTopExtra): one launch less per layer on the protocol-serial chain.  A/B: CM_FRI_TOP_FUSE=0
      if (top_fuse) first_tree.top_extra = MerkleTopExtra{d_chan.u32(), d_alphas.u32(), d_roots.u32()};
      {
        // (round 6) first_leaf_done: the DEEP-quotient kernel of the largest size group has already written the leaf layer
        // (first_tree.leaf_prealloc, adopted by plan_commit): the plan starts at its second launch
        const bool had_leaf = first_leaf_done && first_tree.leaf_prealloc.p != nullptr;
        auto plan = first_tree.plan_commit();
        const int nl = (int)q_logs[0];
        CM_CHECK(!had_leaf || (plan.size() > 1 && plan[0].hi == nl && plan[0].lo == nl && !first_tree.leaf_prealloc.p),
                 "fri: the first-layer tree's plan does not start with a launch of the leaf layer alone");
        for (size_t k = had_leaf ? 1 : 0; k < plan.size(); k++) first_tree.run_launch(plan[k], st);
      }
      if (!first_tree.top_launch_has_extras) chan_mix_root_draw(d_chan.u32(), first_tree.layers[0].u32(), d_alphas.u32(), d_roots.u32(), st);
    } else if (resume->d_chan) {
      CM_CHECK(resume->d_alpha_c != nullptr, "fri: a resumed commit needs both device sources");
      CM_HIP(hipMemcpyAsync(d_chan.p, resume->d_chan, 9 * 4, hipMemcpyDeviceToDevice, st));        // (over the stale upload of P.ch above)
      CM_HIP(hipMemcpyAsync(d_alphas.p, resume->d_alpha_c, 16, hipMemcpyDeviceToDevice, st));   // slot 0 = the circle-fold challenge
    } else {
      uint32_t a4[4];
      resume->alpha_c.to_u32(a4);
      stage_upload(d_alphas.p, a4, sizeof(a4), st);   // slot 0 = the circle-fold challenge, as after a first-layer step
    }
  }
  ColumnSet& layer = last_layer;
  layer = ColumnSet();
  const bool resumed_layer = resume && resume->layer;
  bool layer_is_blank = !pre.empty() && !resumed_layer;   // pre[0] is written (not accumulated into) by the first circle fold: no memset
  if (pre.empty()) {
    if (resumed_layer) layer = std::move(*resume->layer);
    else {
      layer.alloc(std::vector<uint32_t>(4, layer_log), st, false);
      CM_HIP(hipMemsetAsync(layer.buf.p, 0, layer.buf.bytes, st));
    }
  }
  size_t qi = resume ? resume->qi : 0, pi = 0;
  const QM31 unused_alpha;
  // the transcript step of inner layer `li` (1-based slot in alphas / roots), placed into the layer's tree-top launch
  auto set_step = [&](InnerLayer* il, size_t li) {
    if (!top_fuse) return;
    il->tree.top_extra.chan = d_chan.u32();
    il->tree.top_extra.felt_out = d_alphas.u32() + 4 * li;
    il->tree.top_extra.root_log = d_roots.u32() + 8 * li;
  };
  while (layer_log > last_log) {
    if (layer_log <= fri_tail_log()) {
      // every remaining layer in one launch (k_fri_tail); buffers are laid out here so that the decommitment
      // code sees ordinary InnerLayer objects afterwards
      FriTailArgs ta;
      memset(&ta, 0, sizeof(ta));
      ta.tw = view(*P.tw);
      ta.top_log = layer_log; ta.last_log = last_log;
      ta.first_index = (uint32_t)inner.size() + 1;
      ta.chan = d_chan.u32(); ta.alphas = d_alphas.u32(); ta.roots = d_roots.u32();
      for (uint32_t l = layer_log; l > last_log; l--) {
        std::unique_ptr<InnerLayer> il(new InnerLayer());
        il->log = l;
        if (l == layer_log) il->eval = std::move(layer);
        else il->eval.alloc(std::vector<uint32_t>(4, l), st, false);
        FriTailLayer& tl = ta.layers[l];
        for (int c = 0; c < 4; c++) tl.cols[c] = il->eval.ptrs[c];
        while (qi < quotients.size() && q_logs[qi] - 1 == l) {
          CM_CHECK(tl.circle[0] == nullptr, "fri: two quotient groups of one size");
          for (int c = 0; c < 4; c++) tl.circle[c] = quotients[qi].ptrs[c];
          qi++;
        }
        MerkleTree& mt = il->tree;
        mt.cols.assign(il->eval.ptrs.begin(), il->eval.ptrs.end());
        mt.col_logs.assign(4, l);
        mt.layers.resize(l + 1);
        for (uint32_t k = 0; k <= l; k++) { mt.layers[k].alloc((size_t)32 << k); tl.merkle[k] = mt.layers[k].u32(); }
        inner.push_back(std::move(il));
      }
      layer = ColumnSet();
      layer.alloc(std::vector<uint32_t>(4, last_log), st, false);
      for (int c = 0; c < 4; c++) ta.layers[last_log].cols[c] = layer.ptrs[c];
      fri_tail(ta, st);
      layer_log = last_log;
      break;
    }
    // layers above the tail: buffers and tree tables were prepared above (pre[pi])
    InnerLayer* cur = pre[pi].get();
    while (qi < quotients.size() && q_logs[qi] - 1 == layer_log) {
      const uint32_t* src[4] = {quotients[qi].ptrs[0], quotients[qi].ptrs[1], quotients[qi].ptrs[2], quotients[qi].ptrs[3]};
      uint32_t* dst[4] = {cur->eval.ptrs[0], cur->eval.ptrs[1], cur->eval.ptrs[2], cur->eval.ptrs[3]};
      // a blank layer fed by ONE group of quotient columns (the first inner layer): fold and leaf hashes in one pass
      if (layer_is_blank && !cur->planned && !(qi + 1 < quotients.size() && q_logs[qi + 1] - 1 == layer_log)) {
        set_step(cur, inner.size() + 1);
        cur->plan = cur->tree.plan_commit();
        cur->planned = true;
        const int nl = (int)layer_log;
        if (cur->plan.size() > 1 && cur->plan[0].hi == nl && cur->plan[0].lo == nl &&
            fold_circle_leaf(dst, src, q_logs[qi], *P.tw, st, d_alphas.u32(), cur->tree.layers[nl].u32())) {
          cur->leaf_done = true;
          layer_is_blank = false;
          qi++;
          continue;
        }
      }
      fold_circle_into_line(dst, src, q_logs[qi], *P.tw, unused_alpha, !layer_is_blank, st, d_alphas.u32());
      layer_is_blank = false;
      qi++;
    }
    CM_CHECK(!layer_is_blank, "fri: the first layer received no quotient column");
    const size_t li = inner.size() + 1;
    if (!cur->planned) { set_step(cur, li); cur->plan = cur->tree.plan_commit(); cur->planned = true; }
    for (size_t k = cur->leaf_done ? 1 : 0; k < cur->plan.size(); k++) cur->tree.run_launch(cur->plan[k], st);
    if (!cur->tree.top_launch_has_extras) chan_mix_root_draw(d_chan.u32(), cur->tree.layers[0].u32(), d_alphas.u32() + 4 * li, d_roots.u32() + 8 * li, st);
    // fold into the next layer: the next pre-allocated one, or a fresh buffer that the tail / last layer takes over
    uint32_t* dst[4];
    uint32_t* next_leaves = nullptr;   // the leaf layer of the next tree, when that layer is a launch of its own (large layers)
    const bool one_circle_next = qi < quotients.size() && q_logs[qi] == layer_log && !(qi + 1 < quotients.size() && q_logs[qi + 1] == layer_log);
    const bool no_circle_next = !(qi < quotients.size() && q_logs[qi] == layer_log);
    bool fold_in_top = false;   // the next layer is produced by its own tree-top launch (layers of at most 2^MERKLE_TOP_MAX_LOG values)
    if (pi + 1 < pre.size()) {
      InnerLayer* nxt = pre[pi + 1].get();
      for (int c = 0; c < 4; c++) dst[c] = nxt->eval.ptrs[c];
      set_step(nxt, li + 1);
      if (top_fuse && (one_circle_next || no_circle_next)) {
        MerkleTopExtra& x = nxt->tree.top_extra;
        x.fold_mode = one_circle_next ? 2u : 1u;
        const uint32_t R = P.tw->R, L = R - (layer_log + 1);
        x.ixt = P.tw->ixtw + ((1u << (R - 1)) - (1u << (R - 1 - L)));
        x.iyt = P.tw->iytw + (1u << (layer_log - 1));
        x.alpha = d_alphas.u32() + 4 * li;
        x.alpha_c = d_alphas.u32();
        for (int c = 0; c < 4; c++) {
          x.fold_src[c] = cur->eval.ptrs[c];
          x.fold_dst[c] = nxt->eval.ptrs[c];
          x.fold_circ[c] = one_circle_next ? quotients[qi].ptrs[c] : nullptr;
        }
      }
      nxt->plan = nxt->tree.plan_commit();
      nxt->planned = true;
      fold_in_top = nxt->tree.top_launch_has_fold;
      const int nl = (int)layer_log - 1;
      if (nxt->plan.size() > 1 && nxt->plan[0].hi == nl && nxt->plan[0].lo == nl) next_leaves = nxt->tree.layers[nl].u32();
    } else {
      layer = ColumnSet();
      layer.alloc(std::vector<uint32_t>(4, layer_log - 1), st, false);
      for (int c = 0; c < 4; c++) dst[c] = layer.ptrs[c];
    }
    const uint32_t* src[4] = {cur->eval.ptrs[0], cur->eval.ptrs[1], cur->eval.ptrs[2], cur->eval.ptrs[3]};
    // the quotient columns of the next layer's size are folded in by the same kernel (the single-launch tail does its own)
    const bool next_outside_tail = pi + 1 < pre.size();
    if (fold_in_top) {
      if (one_circle_next) qi++;   // folded in by the same launch
    } else if (next_outside_tail && qi < quotients.size() && q_logs[qi] == layer_log &&
        !(qi + 1 < quotients.size() && q_logs[qi + 1] == layer_log)) {
      const uint32_t* circ[4] = {quotients[qi].ptrs[0], quotients[qi].ptrs[1], quotients[qi].ptrs[2], quotients[qi].ptrs[3]};
      if (next_leaves && fold_line_leaf(dst, src, circ, layer_log, *P.tw, st, d_alphas.u32() + 4 * li, d_alphas.u32(), next_leaves))
        pre[pi + 1]->leaf_done = true;
      else fold_line_and_circle(dst, src, circ, layer_log, *P.tw, st, d_alphas.u32() + 4 * li, d_alphas.u32());
      qi++;
    } else {
      // quotient columns of the next layer's size that were NOT folded in by this kernel (two groups of one size: the next
      // iteration accumulates them into the layer) — its leaves must not be hashed before that
      if (qi < quotients.size() && q_logs[qi] == layer_log) next_leaves = nullptr;
      if (next_leaves && fold_line_leaf(dst, src, nullptr, layer_log, *P.tw, st, d_alphas.u32() + 4 * li, nullptr, next_leaves))
        pre[pi + 1]->leaf_done = true;
      else fold_line(dst, src, layer_log, *P.tw, unused_alpha, st, d_alphas.u32() + 4 * li);
    }
    layer_log--;
    inner.push_back(std::move(pre[pi]));
    pi++;
  }
  CM_CHECK(qi == quotients.size(), "fri: not every quotient column was folded");
  d_chan_ = d_chan.u32();
}

// The host's half of the commit phase: challenges, roots and the last layer come back, the transcript steps are replayed and
// cross-checked, the last layer is interpolated (LinePoly) and mixed into the channel.  `from_pinned`: the device-side tail
// (k_tail_last, tail_device.hpp) has already left all three in the calling thread's pinned words and the caller has
// synchronised the stream.
void FriPhase::commit_finish(Prover& P, const cm_pcs_config& cfg, ProofData& pf, bool from_pinned) {
  hipStream_t st = P.st;
  Channel& ch = P.ch;
  const uint32_t n_inner = n_inner_, last_log = last_log_;
  const bool resume = resumed_;
  ColumnSet& layer = last_layer;
  {
    uint32_t n = 1u << last_log;
    const uint32_t* c4[4] = {layer.ptrs[0], layer.ptrs[1], layer.ptrs[2], layer.ptrs[3]};
    std::vector<uint32_t> pos(n);
    for (uint32_t i = 0; i < n; i++) pos[i] = i;
    std::vector<QM31> vals;
    // challenges, roots and the last layer come back in ONE round trip (pinned slots; a large last layer falls back to the
    // batched gather)
    CM_CHECK((n_inner + 1) * 12 <= PIN_LAST_LAYER - PIN_ALPHAS, "fri: too many layers");
    const uint32_t* h_alphas = pinned_words() + PIN_ALPHAS;
    const uint32_t* h_roots = h_alphas + (size_t)(n_inner + 1) * 4;
    if (!from_pinned) CM_HIP(hipMemcpyAsync((void*)h_alphas, d_ar.p, (size_t)(n_inner + 1) * 48, hipMemcpyDeviceToHost, st));
    if (from_pinned || 4 * n <= PIN_WORDS - PIN_LAST_LAYER) {
      CM_CHECK(4 * n <= PIN_WORDS - PIN_LAST_LAYER, "fri: last layer does not fit the pinned slot");
      uint32_t* ll = pinned_words() + PIN_LAST_LAYER;
      if (!from_pinned) {
        if (c4[1] == c4[0] + n && c4[2] == c4[0] + 2 * n && c4[3] == c4[0] + 3 * n) CM_HIP(hipMemcpyAsync(ll, c4[0], n * 16, hipMemcpyDeviceToHost, st));   // one arena
        else for (int k = 0; k < 4; k++) CM_HIP(hipMemcpyAsync(ll + k * n, c4[k], n * 4, hipMemcpyDeviceToHost, st));
        CM_HIP(hipStreamSynchronize(st));
      }
      for (uint32_t i = 0; i < n; i++) {
        uint32_t w4[4] = {ll[i], ll[n + i], ll[2 * n + i], ll[3 * n + i]};
        vals.push_back(QM31::from_u32(w4));
      }
    } else {
      GatherBatch gb;
      QGather g = plan_gather_q(c4, pos, gb);
      gb.run(st);  // synchronises the stream: roots / challenges are on the host now
      finish_gather_q(g, gb, vals);
    }
    // host replay of the device-side transcript steps
    CM_CHECK(inner.size() == n_inner, "fri: layer count mismatch");
    for (size_t li = resume ? 1 : 0; li <= n_inner; li++) {
      hostch::Hash32 root;
      memcpy(root.data(), &h_roots[8 * li], 32);
      ch.mix_root(root);
      QM31 alpha = ch.draw_felt();
      CM_CHECK(alpha == QM31::from_u32(&h_alphas[4 * li]), "fri: device transcript diverged from the host channel");
      if (li == 0) pf.fri_first.commitment = root;
      else inner[li - 1]->root = root;
    }
    for (uint32_t l = 0; l < last_log; l++) {
      uint32_t stride = 1u << l;
      for (uint32_t h = 0; h < (n >> (l + 1)); h++) {
        // LineDomain(half_odds(last_log)) doubled l times: coset half_odds(last_log - l); point bitrev(h)
        uint32_t clog_ = last_log - l;
        uint32_t idx = subgroup_gen_index(clog_ + 2) + subgroup_gen_index(clog_) * bit_reverse(h, clog_ - 1);
        M31 xinv = inv(point_at_index(idx).x);
        for (uint32_t k = 0; k < stride; k++) {
          uint32_t i0 = (h << (l + 1)) + k, i1 = i0 + stride;
          QM31 a = vals[i0], b = vals[i1];
          vals[i0] = a + b;
          vals[i1] = (a - b) * xinv;
        }
      }
    }
    // `vals` sits in bit-reversed evaluation order, so after the in-place transform position p holds the coefficient of
    // the basis element of degree p (= LinePoly::into_ordered_coefficients); the proof keeps the first 2^bound of them
    // in LinePoly's own bit-reversed order (from_ordered_coefficients).
    M31 ninv = inv(M31::from_u32(n));
    uint32_t keep = 1u << cfg.log_last_layer_degree_bound;
    for (uint32_t i = keep; i < n; i++) CM_CHECK(vals[i].is_zero(), "fri: last layer has invalid degree");
    pf.last_layer_poly.assign(keep, QM31());
    for (uint32_t i = 0; i < keep; i++) pf.last_layer_poly[bit_reverse(i, cfg.log_last_layer_degree_bound)] = vals[i] * ninv;
    pf.last_layer_log_size = cfg.log_last_layer_degree_bound;
    ch.mix_felts(pf.last_layer_poly.data(), pf.last_layer_poly.size());
  }
  d_ar.release();
}

}  // namespace cm
