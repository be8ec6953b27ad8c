"""rand 0.8 `StdRng::seed_from_u64` restated in Python (TEST INFRASTRUCTURE, unverified).

The reference's tests and benches draw all inputs from `StdRng::seed_from_u64(seed)`
(tests/schnorr.rs:16, benches/signature.rs:81).  Restated from the published behaviour of
rand_core 0.6 / rand_chacha 0.3:
  * seed_from_u64: PCG32 (multiplier 6364136223846793005, increment 11634580027462260723; the
    state is advanced first, output = rotate_right(((s >> 18) ^ s) >> 27, s >> 59)) emits eight
    u32, little-endian, = the 32-byte ChaCha key;
  * StdRng = ChaCha12Rng: 12-round ChaCha, 64-bit block counter (words 12, 13) starting at 0,
    64-bit stream id (words 14, 15) = 0; `fill_bytes` hands out the keystream in order (whole u32
    words; our draws are 64 bytes each, so alignment never matters).
Nothing in /root/reference pins these bytes; they become checkable the day someone runs
rust/dusk-schnorr-gpu/src/bin/golden_gen.rs.
"""
import struct

MASK32 = 0xFFFFFFFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & MASK32


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key_words, counter, stream=0, rounds=12):
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & MASK32, (counter >> 32) & MASK32, stream & MASK32, (stream >> 32) & MASK32]
    s = list(init)
    for _ in range(rounds // 2):
        _qr(s, 0, 4, 8, 12); _qr(s, 1, 5, 9, 13); _qr(s, 2, 6, 10, 14); _qr(s, 3, 7, 11, 15)
        _qr(s, 0, 5, 10, 15); _qr(s, 1, 6, 11, 12); _qr(s, 2, 7, 8, 13); _qr(s, 3, 4, 9, 14)
    return struct.pack("<16I", *[(x + y) & MASK32 for x, y in zip(s, init)])


def seed_from_u64(state):
    MUL, INC = 6364136223846793005, 11634580027462260723
    words = []
    for _ in range(8):
        state = (state * MUL + INC) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((state >> 18) ^ state) >> 27) & MASK32
        rot = state >> 59
        words.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & MASK32)
    return words


class StdRng:
    def __init__(self, seed_u64):
        self.key = seed_from_u64(seed_u64)
        self.counter = 0
        self.buf = b""

    def fill_bytes(self, n):
        while len(self.buf) < n:
            self.buf += chacha_block(self.key, self.counter)
            self.counter += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out


def chacha20_rfc7539_selftest():
    """RFC 7539 §2.3.2 block-function vector (20 rounds, 32-bit counter + 96-bit nonce laid out in
    the same four words) — checks the quarter round / word order of this file, not rand's use."""
    key = list(struct.unpack("<8I", bytes(range(32))))
    # counter = 1, nonce = 00 00 00 09 00 00 00 4a 00 00 00 00  -> words 12..15
    init_tail = [1, 0x09000000, 0x4A000000, 0]
    counter = init_tail[0] | (init_tail[1] << 32)
    stream = init_tail[2] | (init_tail[3] << 32)
    out = chacha_block(key, counter, stream, rounds=20)
    return out[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"
