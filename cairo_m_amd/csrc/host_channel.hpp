// Host side of the Fiat-Shamir transcript for the HIP prover: Blake2s (RFC 7693) and Stwo's
// Blake2sChannel framing.  Reference call sites: crates/prover/src/prover.rs:33-36, 66, 73, 78, 90-91, 94,
// 98 and the channel use inside stwo `prove` (prover.rs:131).  The transcript is tiny (a few hundred
// hashes per proof) and inherently sequential, so it stays on the host; the heavy hashing (Merkle
// layers, proof-of-work search) runs in kernels_hash.hip.
#pragma once
#include <algorithm>
#include <stdint.h>
#include <string.h>
#include <array>
#include <vector>
#include "field.hpp"
#include "framing.hpp"

namespace cm {
namespace hostch {

using Hash32 = std::array<uint8_t, 32>;

// Transcript log (cm_set_transcript_log / cm_proof_transcript): one entry per Channel-trait call the reference prover makes
// (Stwo `Channel::{mix_u32s, mix_felts, mix_u64, draw_felt, draw_felts, draw_random_bytes}` + `MerkleChannel::mix_root`), with
// the digest AFTER the call.  integration/prover-hip/tests/golden_dump.rs records the same entries from the reference through
// a logging channel wrapper, and the oracle records them too: tests compare the three step by step.
struct TranscriptEntry {
  const char* op;
  Hash32 digest;                 // channel digest after the call
  uint32_t n_words;              // words mixed in / drawn
  std::vector<uint32_t> words;   // the first <= 16 of them (mixed input or drawn output)
};
using TranscriptLog = std::vector<TranscriptEntry>;
// a copied Channel (PoW probes) must not write into the log of the channel it was copied from
struct LogRef {
  TranscriptLog* p = nullptr;
  LogRef() {}
  LogRef(const LogRef&) : p(nullptr) {}
  LogRef& operator=(const LogRef&) { return *this; }
};

#if defined(__x86_64__) && defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(CM_HOST_B2S_SCALAR)
}  // namespace hostch
}  // namespace cm
#include <immintrin.h>
namespace cm {
namespace hostch {
// AVX-512VL form (EPYC Zen 4 / 5 and Xeon hosts; chosen at run time): the sixteen message words sit in ONE zmm register and a
// round's permutation is one vpermd; every rotation is one vprord instead of shift + shift + or.  The host hashes ~30 KB of
// sampled values into the transcript with the GPU waiting for the random coefficient that follows (mix_felts): ~100 -> ~55 us.
__attribute__((target("avx512f,avx512vl"))) inline void compress_avx512(uint32_t h[8], const uint32_t m[16], uint64_t t, uint32_t f0) {
  alignas(64) static const uint32_t IDX[10][16] = {
#define CM_HB_IDX(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  {s0, s2, s4, s6, s1, s3, s5, s7, s8, s10, s12, s14, s9, s11, s13, s15}
      CM_HB_IDX(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), CM_HB_IDX(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3),
      CM_HB_IDX(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4), CM_HB_IDX(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8),
      CM_HB_IDX(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13), CM_HB_IDX(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9),
      CM_HB_IDX(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11), CM_HB_IDX(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10),
      CM_HB_IDX(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5), CM_HB_IDX(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)};
#undef CM_HB_IDX
  const __m128i h0 = _mm_loadu_si128((const __m128i*)h), h1 = _mm_loadu_si128((const __m128i*)(h + 4));
  __m128i a = h0, b = h1;
  __m128i c = _mm_set_epi32((int)0xA54FF53Au, (int)0x3C6EF372u, (int)0xBB67AE85u, (int)0x6A09E667u);
  __m128i d = _mm_set_epi32((int)0x5BE0CD19u, (int)(0x1F83D9ABu ^ f0), (int)(0x9B05688Cu ^ (uint32_t)(t >> 32)), (int)(0x510E527Fu ^ (uint32_t)t));
  const __m512i mv = _mm512_loadu_si512((const void*)m);
#define CM_HB_G5(x, y)                                                                           \
  a = _mm_add_epi32(_mm_add_epi32(a, b), x); d = _mm_ror_epi32(_mm_xor_si128(d, a), 16);         \
  c = _mm_add_epi32(c, d); b = _mm_ror_epi32(_mm_xor_si128(b, c), 12);                           \
  a = _mm_add_epi32(_mm_add_epi32(a, b), y); d = _mm_ror_epi32(_mm_xor_si128(d, a), 8);          \
  c = _mm_add_epi32(c, d); b = _mm_ror_epi32(_mm_xor_si128(b, c), 7);
  for (int r = 0; r < 10; r++) {
    const __m512i p = _mm512_permutexvar_epi32(_mm512_load_si512((const void*)IDX[r]), mv);
    const __m128i x1 = _mm512_castsi512_si128(p), y1 = _mm512_extracti32x4_epi32(p, 1);
    const __m128i x2 = _mm512_extracti32x4_epi32(p, 2), y2 = _mm512_extracti32x4_epi32(p, 3);
    CM_HB_G5(x1, y1)
    b = _mm_shuffle_epi32(b, _MM_SHUFFLE(0, 3, 2, 1));   // diagonals: lane i takes b[i+1], c[i+2], d[i+3]
    c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
    d = _mm_shuffle_epi32(d, _MM_SHUFFLE(2, 1, 0, 3));
    CM_HB_G5(x2, y2)
    b = _mm_shuffle_epi32(b, _MM_SHUFFLE(2, 1, 0, 3));
    c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
    d = _mm_shuffle_epi32(d, _MM_SHUFFLE(0, 3, 2, 1));
  }
#undef CM_HB_G5
  _mm_storeu_si128((__m128i*)h, _mm_xor_si128(h0, _mm_xor_si128(a, c)));
  _mm_storeu_si128((__m128i*)(h + 4), _mm_xor_si128(h1, _mm_xor_si128(b, d)));
}
inline bool host_has_avx512vl() {
  static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && getenv("CM_HOST_B2S_NO_AVX512") == nullptr;
  return v;
}
// The four columns (then the four diagonals) of a round in one 128-bit register each (SSE2, part of every x86-64): the host mixes
// ~20 KB of sampled values into the transcript with the GPU idle (mix_felts: 54 -> ~30 us), and the verifier hashes every
// decommitted node.  Same function as the scalar form below (CM_HOST_B2S_SCALAR builds that one).
inline void compress(uint32_t h[8], const uint32_t m[16], uint64_t t, uint32_t f0) {
  if (host_has_avx512vl()) { compress_avx512(h, m, t, f0); return; }
  static const uint8_t S[10][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
  const __m128i h0 = _mm_loadu_si128((const __m128i*)h), h1 = _mm_loadu_si128((const __m128i*)(h + 4));
  __m128i a = h0, b = h1;
  __m128i c = _mm_set_epi32((int)0xA54FF53Au, (int)0x3C6EF372u, (int)0xBB67AE85u, (int)0x6A09E667u);
  __m128i d = _mm_set_epi32((int)0x5BE0CD19u, (int)(0x1F83D9ABu ^ f0), (int)(0x9B05688Cu ^ (uint32_t)(t >> 32)), (int)(0x510E527Fu ^ (uint32_t)t));
#define CM_HB_ROT(x, n) _mm_or_si128(_mm_srli_epi32(x, n), _mm_slli_epi32(x, 32 - (n)))
#define CM_HB_G(x, y)                                                                  \
  a = _mm_add_epi32(_mm_add_epi32(a, b), x); d = _mm_xor_si128(d, a); d = CM_HB_ROT(d, 16); \
  c = _mm_add_epi32(c, d); b = _mm_xor_si128(b, c); b = CM_HB_ROT(b, 12);                  \
  a = _mm_add_epi32(_mm_add_epi32(a, b), y); d = _mm_xor_si128(d, a); d = CM_HB_ROT(d, 8);  \
  c = _mm_add_epi32(c, d); b = _mm_xor_si128(b, c); b = CM_HB_ROT(b, 7);
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = S[r];
    __m128i x = _mm_set_epi32((int)m[s[6]], (int)m[s[4]], (int)m[s[2]], (int)m[s[0]]);
    __m128i y = _mm_set_epi32((int)m[s[7]], (int)m[s[5]], (int)m[s[3]], (int)m[s[1]]);
    CM_HB_G(x, y)
    b = _mm_shuffle_epi32(b, _MM_SHUFFLE(0, 3, 2, 1));   // diagonals: lane i takes b[i+1], c[i+2], d[i+3]
    c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
    d = _mm_shuffle_epi32(d, _MM_SHUFFLE(2, 1, 0, 3));
    x = _mm_set_epi32((int)m[s[14]], (int)m[s[12]], (int)m[s[10]], (int)m[s[8]]);
    y = _mm_set_epi32((int)m[s[15]], (int)m[s[13]], (int)m[s[11]], (int)m[s[9]]);
    CM_HB_G(x, y)
    b = _mm_shuffle_epi32(b, _MM_SHUFFLE(2, 1, 0, 3));
    c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
    d = _mm_shuffle_epi32(d, _MM_SHUFFLE(0, 3, 2, 1));
  }
#undef CM_HB_G
#undef CM_HB_ROT
  _mm_storeu_si128((__m128i*)h, _mm_xor_si128(h0, _mm_xor_si128(a, c)));
  _mm_storeu_si128((__m128i*)(h + 4), _mm_xor_si128(h1, _mm_xor_si128(b, d)));
}
#else
inline void compress(uint32_t h[8], const uint32_t m[16], uint64_t t, uint32_t f0) {
  static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  static const uint8_t S[10][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
  uint32_t v[16];
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[8 + i] = IV[i]; }
  v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32); v[14] ^= f0;
  auto rot = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
    v[a] += v[b] + x; v[d] = rot(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rot(v[b] ^ v[c], 12);
    v[a] += v[b] + y; v[d] = rot(v[d] ^ v[a], 8);  v[c] += v[d]; v[b] = rot(v[b] ^ v[c], 7);
  };
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = S[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]);
    G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
    G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]);
    G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
}
#endif
inline Hash32 blake2s256(const uint8_t* data, size_t len) {
  uint32_t h[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
  uint64_t t = 0;
  size_t off = 0;
  uint32_t m[16];
  while (len - off > 64) {
    memcpy(m, data + off, 64);
    t += 64;
    compress(h, m, t, 0);
    off += 64;
  }
  uint8_t block[64] = {0};
  size_t rem = len - off;
  if (rem) memcpy(block, data + off, rem);
  t += rem;
  memcpy(m, block, 64);
  compress(h, m, t, 0xFFFFFFFFu);
  Hash32 out;
  memcpy(out.data(), h, 32);
  return out;
}

// Blake2s-256 over a message that arrives in pieces (same digest as blake2s256 over the concatenation)
struct Blake2sStream {
  uint32_t h[8];
  uint64_t t = 0;
  uint8_t buf[64];
  size_t fill = 0;
  Blake2sStream() {
    static const uint32_t iv[8] = {0x6A09E667u ^ 0x01010020u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    memcpy(h, iv, 32);
  }
  void update(const void* data, size_t len) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t m[16];
    while (len) {
      if (fill == 64) {   // a full buffer is only compressed once more input follows (the last block carries the final flag)
        memcpy(m, buf, 64);
        t += 64;
        compress(h, m, t, 0);
        fill = 0;
      }
      const size_t take = std::min(len, 64 - fill);
      memcpy(buf + fill, p, take);
      fill += take; p += take; len -= take;
    }
  }
  Hash32 finish() {
    uint32_t m[16];
    if (fill < 64) memset(buf + fill, 0, 64 - fill);
    memcpy(m, buf, 64);
    t += fill;
    compress(h, m, t, 0xFFFFFFFFu);
    Hash32 out;
    memcpy(out.data(), h, 32);
    return out;
  }
};

struct Channel {
  Hash32 digest{};
  uint32_t n_challenges = 0, n_sent = 0;
  LogRef log;
  void note(const char* op, const uint32_t* w, size_t n) {
    if (!log.p) return;
    TranscriptEntry e{op, digest, (uint32_t)n, {}};
    e.words.assign(w, w + (n < 16 ? n : 16));
    log.p->push_back(std::move(e));
  }
  void update(const Hash32& d) { digest = d; n_challenges++; n_sent = 0; }
  uint32_t trailing_zeros() const {
    uint32_t w[4];
    memcpy(w, digest.data(), 16);
    for (int i = 0; i < 4; i++) if (w[i]) return 32 * i + __builtin_ctz(w[i]);
    return 128;
  }
  void absorb_u32s(const uint32_t* w, size_t n) {
    std::vector<uint8_t> buf(32 + 4 * n);
    memcpy(buf.data(), digest.data(), 32);
    if (n) memcpy(buf.data() + 32, w, 4 * n);
    update(blake2s256(buf.data(), buf.size()));
  }
  void mix_u32s(const uint32_t* w, size_t n) {
    absorb_u32s(w, n);
    note("mix_u32s", w, n);
  }
  void mix_felts(const QM31* f, size_t n) {
    std::vector<uint32_t> w(4 * n);
    for (size_t i = 0; i < n; i++) f[i].to_u32(&w[4 * i]);
    absorb_u32s(w.data(), w.size());
    note("mix_felts", w.data(), w.size());
  }
  // mix_felts over felts that arrive in pieces (the OODS values come back from the GPU in two parts and the first is hashed
  // while the second is still being computed): begin, update ..., end == one mix_felts over the concatenation
  struct FeltMixer { Blake2sStream st; size_t n_words = 0; uint32_t first[16]; };
  void mix_felts_begin(FeltMixer& fm) const { fm.st.update(digest.data(), 32); }
  void mix_felts_update(FeltMixer& fm, const QM31* f, size_t n) const {
    uint32_t w[64];
    for (size_t i = 0; i < n;) {
      const size_t c = std::min<size_t>(16, n - i);
      for (size_t k = 0; k < c; k++) f[i + k].to_u32(&w[4 * k]);
      for (size_t k = 0; k < 4 * c && fm.n_words + k < 16; k++) fm.first[fm.n_words + k] = w[k];
      fm.st.update(w, 16 * c);
      fm.n_words += 4 * c;
      i += c;
    }
  }
  void mix_felts_end(FeltMixer& fm) {
    update(fm.st.finish());
    if (log.p) {
      TranscriptEntry e{"mix_felts", digest, (uint32_t)fm.n_words, {}};
      e.words.assign(fm.first, fm.first + (fm.n_words < 16 ? fm.n_words : 16));
      log.p->push_back(std::move(e));
    }
  }
  // framing switch `mix_u64` (framing.hpp): raw compression F(digest, [lo, hi, 0...], t=0, f=0) — the form Stwo's SIMD grind
  // searches over — or mix_u32s(&[lo, hi])
  void mix_u64(uint64_t v) {
    const uint32_t lohi[2] = {(uint32_t)v, (uint32_t)(v >> 32)};
    if (framing().mix_u64_u32s) {
      absorb_u32s(lohi, 2);
    } else {
      uint32_t h[8], m[16] = {0};
      memcpy(h, digest.data(), 32);
      m[0] = lohi[0]; m[1] = lohi[1];
      compress(h, m, 0, 0);
      Hash32 d;
      memcpy(d.data(), h, 32);
      update(d);
    }
    note("mix_u64", lohi, 2);
  }
  void mix_root(const Hash32& root) {
    uint8_t buf[64];
    memcpy(buf, digest.data(), 32);
    memcpy(buf + 32, root.data(), 32);
    update(blake2s256(buf, 64));
    uint32_t w[8];
    memcpy(w, root.data(), 32);
    note("mix_root", w, 8);
  }
  Hash32 random_bytes() {   // unlogged core of draw_random_bytes
    uint8_t buf[65] = {0};
    memcpy(buf, digest.data(), 32);
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
      for (int i = 0; i < 8; i++) ok = ok && u[i] < 2 * P;
      if (!ok) continue;
      for (int i = 0; i < 8; i++) out[i] = M31::from_u32(u[i]);
      return;
    }
  }
  QM31 draw_felt() {
    M31 f[8];
    draw_base_felts(f);
    const uint32_t w[4] = {f[0].v, f[1].v, f[2].v, f[3].v};
    note("draw_felt", w, 4);
    return QM31(f[0], f[1], f[2], f[3]);
  }
  void draw_two_felts(QM31& a, QM31& b) {  // draw_felts(2): one hash, 8 base felts
    M31 f[8];
    draw_base_felts(f);
    a = QM31(f[0], f[1], f[2], f[3]);
    b = QM31(f[4], f[5], f[6], f[7]);
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = f[i].v;
    note("draw_felts", w, 8);
  }
};

}  // namespace hostch
}  // namespace cm
