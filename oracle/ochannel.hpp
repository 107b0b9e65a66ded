// ORACLE (test infrastructure).  Blake2sChannel (Fiat-Shamir transcript) + proof-of-work grind.
// Restates Stwo `core::channel::blake2s` and `core::proof_of_work` (PARITY UNPINNED — Stwo not
// vendored).  In-tree call sites that pin the *use*: `pcs_config.mix_into(channel)`
// (crates/prover/src/prover.rs:36), `channel.mix_u32s` (public_data.rs:401-411),
// `channel.mix_u64(log_size)` (components/opcodes/store_fp_imm.rs:124),
// `channel.mix_felts(&[claimed_sum])` (store_fp_imm.rs:304), the PoW predicate
// `channel.mix_u64(nonce); channel.trailing_zeros() >= bits` (verifier.rs:55-58).
//
// Framing choices (each behind one function so it can be corrected in one place):
//   mix_u64   : raw compression  digest' = F(digest, [lo,hi,0..], 0,0,0,0)   (framing switch mix_u64=raw, oframing.hpp)
//               — the form that makes Stwo's SIMD `grind` (compress16 over nonces) agree with
//               `mix_u64` + `trailing_zeros`; alternative (mix_u64=u32s) = mix_u32s([lo,hi]).
//   mix_u32s / mix_felts : digest' = Blake2s256(digest || LE words)
//   draw_random_bytes    : Blake2s256(digest || LE32(n_sent) zero-padded to 32 B || 0x00)
#pragma once
#include "oblake2s.hpp"
#include "ofield.hpp"
#include "oframing.hpp"

namespace orc {

// Transcript log: one entry per Channel-trait call of the reference prover (op name, digest after the call, the words mixed
// in / drawn) — the same records the product (cm_proof_transcript) and the reference harness
// (integration/prover-hip/tests/golden_dump.rs) produce; compared step by step in tests/.
struct TranscriptEntry { const char* op; Hash32 digest; uint32_t n_words; std::vector<uint32_t> words; };
using TranscriptLog = std::vector<TranscriptEntry>;
struct LogRef {   // a copied Channel (grind probes) does not log
  TranscriptLog* p = nullptr;
  LogRef() {}
  LogRef(const LogRef&) : p(nullptr) {}
  LogRef& operator=(const LogRef&) { return *this; }
};

struct Channel {
  Hash32 digest{};  // all-zero default
  uint32_t n_challenges = 0, n_sent = 0;
  LogRef log;
  void note(const char* op, const uint32_t* w, size_t n) {
    if (!log.p) return;
    TranscriptEntry e{op, digest, (uint32_t)n, {}};
    e.words.assign(w, w + (n < 16 ? n : 16));
    log.p->push_back(std::move(e));
  }

  void update_digest(const Hash32& d) {
    digest = d;
    n_challenges++;
    n_sent = 0;
  }
  uint32_t trailing_zeros() const {
    // u128 from the first 16 bytes, little endian
    uint32_t tz = 0;
    for (int i = 0; i < 16; i++) {
      uint8_t b = digest[i];
      if (b == 0) { tz += 8; continue; }
      while (!(b & 1)) { tz++; b >>= 1; }
      return tz;
    }
    return 128;
  }
  void absorb_u32s(const uint32_t* w, size_t n) {
    std::vector<uint8_t> buf(32 + 4 * n);
    memcpy(buf.data(), digest.data(), 32);
    if (n) memcpy(buf.data() + 32, w, 4 * n);
    update_digest(blake2s256(buf));
  }
  void mix_u32s(const uint32_t* w, size_t n) {
    absorb_u32s(w, n);
    note("mix_u32s", w, n);
  }
  void mix_u32s(const std::vector<uint32_t>& w) { mix_u32s(w.data(), w.size()); }
  void mix_felts(const QM31* f, size_t n) {
    std::vector<uint32_t> w(4 * n);
    for (size_t i = 0; i < n; i++) f[i].to_u32(&w[4 * i]);
    absorb_u32s(w.data(), w.size());
    note("mix_felts", w.data(), w.size());
  }
  void mix_felts(const std::vector<QM31>& f) { mix_felts(f.data(), f.size()); }
  void mix_u64(uint64_t v) {
    const uint32_t w[2] = {(uint32_t)v, (uint32_t)(v >> 32)};
    if (framing().mix_u64_u32s) {   // framing switch mix_u64=u32s
      absorb_u32s(w, 2);
    } else {                        // mix_u64=raw (default): one raw compression
      uint32_t h[8], m[16] = {0};
      memcpy(h, digest.data(), 32);
      m[0] = w[0];
      m[1] = w[1];
      b2s_compress(h, m, 0, 0, 0, 0);
      Hash32 d;
      memcpy(d.data(), h, 32);
      update_digest(d);
    }
    note("mix_u64", w, 2);
  }
  void mix_root(const Hash32& root) {  // Blake2sMerkleChannel::mix_root
    uint8_t buf[64];
    memcpy(buf, digest.data(), 32);
    memcpy(buf + 32, root.data(), 32);
    update_digest(blake2s256(buf, 64));
    uint32_t w[8];
    memcpy(w, root.data(), 32);
    note("mix_root", w, 8);
  }
  Hash32 random_bytes() {   // the unlogged core of draw_random_bytes
    uint8_t buf[65];
    memcpy(buf, digest.data(), 32);
    memset(buf + 32, 0, 33);
    memcpy(buf + 32, &n_sent, 4);
    n_sent++;
    return blake2s256(buf, 65);
  }
  Hash32 draw_random_bytes() {
    Hash32 b = random_bytes();
    uint32_t w[8];
    memcpy(w, b.data(), 32);
    note("draw_random_bytes", w, 8);
    return b;
  }
  void draw_base_felts(M31 out[8]) {
    for (;;) {
      Hash32 b = random_bytes();
      uint32_t u[8];
      memcpy(u, b.data(), 32);
      bool ok = true;
      for (int i = 0; i < 8; i++) ok = ok && (u[i] < 2 * P);
      if (!ok) continue;
      for (int i = 0; i < 8; i++) out[i] = M31(u[i]);
      return;
    }
  }
  QM31 draw_felt() {
    M31 f[8];
    draw_base_felts(f);
    QM31 r = QM31::from_m31s(f[0], f[1], f[2], f[3]);
    uint32_t w[4];
    r.to_u32(w);
    note("draw_felt", w, 4);
    return r;
  }
  std::vector<QM31> draw_felts(size_t n) {
    std::vector<QM31> out;
    M31 f[8];
    size_t have = 0;
    while (out.size() < n) {
      if (have == 0) { draw_base_felts(f); have = 8; }
      size_t o = 8 - have;
      out.push_back(QM31::from_m31s(f[o], f[o + 1], f[o + 2], f[o + 3]));
      have -= 4;
    }
    std::vector<uint32_t> w(4 * n);
    for (size_t i = 0; i < n; i++) out[i].to_u32(&w[4 * i]);
    note("draw_felts", w.data(), w.size());
    return out;
  }
};

// Smallest nonce such that mix_u64(nonce) leaves >= pow_bits trailing zeros.
inline uint64_t grind(const Channel& ch, uint32_t pow_bits) {
  for (uint64_t nonce = 0;; nonce++) {
    Channel c = ch;
    c.mix_u64(nonce);
    if (c.trailing_zeros() >= pow_bits) return nonce;
  }
}

}  // namespace orc
