// cd_perm.hpp -- the per-sweep visiting order of the coordinate-descent solver.
//
// The reference reshuffles the active list before every sweep with a libc
// rand() swap shuffle (src/libslim/cd.c:76-86, :115); the sequence is neither
// seeded nor reproducible across threads, and any order is a valid CD order.
// The engine instead walks a *stateless keyed permutation* of [0,n): position
// p of sweep t of item iC visits active slot perm(p; key(seed,iC,t)).  No
// memory traffic, no per-wave RNG state, O(1) per visit on the scalar unit,
// and the CPU oracle can walk the very same order (oracle/slim_oracle.c holds
// an independent integer-exact twin), so engine-vs-oracle parity can be
// tested visit for visit.
//
// perm is a bijection on b = ceil(log2 n) bits -- three rounds of (odd
// multiply + add mod 2^b, xorshift right), each invertible -- followed by
// cycle walking into [0,n) (expected < 2 rounds because 2^b < 2n).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define SLIM_HD __host__ __device__ __forceinline__
#else
#define SLIM_HD inline
#endif

namespace slimamd {

SLIM_HD uint32_t perm_key(uint32_t seed, uint32_t item, uint32_t sweep) {
  uint32_t h = seed * 0x9E3779B1u + 0x7F4A7C15u;
  h ^= item + 0x85EBCA6Bu + (h << 6) + (h >> 2);
  h *= 0xC2B2AE35u;
  h ^= h >> 15;
  h ^= sweep * 0x27D4EB2Fu + 0x165667B1u + (h << 6) + (h >> 2);
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

struct PermCtx {
  uint32_t n, mask, s1, s2;
  uint32_t a1, c1, a2, c2, a3, c3;
};

SLIM_HD PermCtx perm_make(uint32_t n, uint32_t key) {
  PermCtx c;
  c.n = n;
  uint32_t b = n > 1 ? 32u - (uint32_t)__builtin_clz(n - 1) : 1u;
  c.mask = b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u);
  c.s1 = (b + 1) / 2;
  c.s2 = (b + 2) / 3;
  if (c.s2 == 0) c.s2 = 1;
  c.a1 = key | 1u;
  c.c1 = key >> 7;
  c.a2 = ((key * 0x9E3779B1u) >> 3) | 1u;
  c.c2 = (key * 0x85EBCA6Bu) >> 11;
  c.a3 = ((key * 0xC2B2AE35u) >> 5) | 1u;
  c.c3 = key >> 17;
  return c;
}

SLIM_HD uint32_t perm_index(const PermCtx& c, uint32_t p) {
  if (c.n <= 1) return 0;
  uint32_t x = p;
  do {
    x = (x * c.a1 + c.c1) & c.mask;
    x ^= x >> c.s1;
    x = (x * c.a2 + c.c2) & c.mask;
    x ^= x >> c.s2;
    x = (x * c.a3 + c.c3) & c.mask;
    x ^= x >> c.s1;
  } while (x >= c.n);
  return x;
}

}  // namespace slimamd
