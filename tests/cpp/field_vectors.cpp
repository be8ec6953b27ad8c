// Prints host-arithmetic vectors of include/dusk_schnorr.hpp's BlsScalar / JubJubScalar (the
// reference's in-memory representation: Montgomery limbs, R = 2^256) for tests/test_cpp_mirror.py,
// which re-computes every line with Python integers.  No GPU call is made.
//   <field> <a> <b> <a*b> <a+b> <a-b> <-a> <1/a> <from_bytes_wide(w)> <limbs of a>      (hex, canonical LE)
#include <cstdio>

#include "dusk_schnorr.hpp"

using namespace dusk_schnorr;

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  void operator()(uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(next() >> 32);
  }
};
static void hex(const uint8_t* b, size_t n) {
  for (size_t i = 0; i < n; i++) std::printf("%02x", b[i]);
  std::printf(" ");
}
template <class S>
static void lines(const char* name, Rng& rng, int count) {
  for (int k = 0; k < count; k++) {
    uint8_t w[64];
    rng(w, 64);
    if (k == 0) std::memset(w, 0xff, 64);            // the largest wide value
    if (k == 1) std::memset(w, 0, 64), w[0] = 1;     // one
    const S a = S::random(rng), b = k == 2 ? -S::one() : S::random(rng);
    std::printf("%s ", name);
    hex(a.to_bytes().data(), 32);
    hex(b.to_bytes().data(), 32);
    hex((a * b).to_bytes().data(), 32);
    hex((a + b).to_bytes().data(), 32);
    hex((a - b).to_bytes().data(), 32);
    hex((-a).to_bytes().data(), 32);
    hex(a.invert()->to_bytes().data(), 32);
    hex(w, 64);
    hex(S::from_bytes_wide(w).to_bytes().data(), 32);
    hex(reinterpret_cast<const uint8_t*>(a.l), 32);
    std::printf("\n");
  }
}
int main() {
  Rng rng{2321};
  lines<BlsScalar>("fq", rng, 40);
  lines<JubJubScalar>("fr", rng, 40);
  return 0;
}
