"""Parser for fixtures produced by the REAL dusk-schnorr 0.18 — TEST INFRASTRUCTURE.

Nothing under /root/reference pins the challenge hash; the reference cannot be
built in the authoring image.  A maintainer with cargo runs
    cargo run --bin golden_gen > tests/golden/reference_vectors.txt      (rust/dusk-schnorr-gpu)
and drops the file in; from then on tests/test_oracle.py::test_reference_fixtures_pin_the_oracle
and tests/test_gpu_parity.py::test_reference_fixtures_on_gpu compare the oracle and the HIP engine
with it and parity is pinned without any code change.  Any file tests/golden/reference_*.txt in
golden_gen.rs's line format is picked up; tests/golden/reference_*.json (a list of the same
records as dicts) as well.

Line format (one record per line, fields separated by blanks, hex = little-endian to_bytes()):
    sponge_hash_1_2_3_le <hex32>
    truncated_hash_1_2_3_le <hex32>
    sig <i> sk <hex32> m <hex32> u <hex32> R <hex64> PK <hex64> sig_bytes <hex64> pk_bytes <hex32> verdict <true|false>
    from_bytes <i> <hex32> ok u <hex32> v <hex32>
    from_bytes <i> <hex32> err
"""
import glob
import json
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _parse_line(line):
    t = line.split()
    if not t:
        return None
    if t[0] in ("sponge_hash_1_2_3_le", "truncated_hash_1_2_3_le"):
        return {"kind": t[0], "hex": t[1]}
    if t[0] == "sig":
        rec = {"kind": "sig", "i": int(t[1])}
        for k, v in zip(t[2::2], t[3::2]):
            rec[k] = v
        rec["verdict"] = rec["verdict"] == "true"
        return rec
    if t[0] == "from_bytes":
        rec = {"kind": "from_bytes", "i": int(t[1]), "enc": t[2], "ok": t[3] == "ok"}
        if rec["ok"]:
            rec["u"], rec["v"] = t[5], t[7]
        return rec
    raise ValueError("unknown fixture line: %r" % line[:60])


def load():
    """All records of every dropped-in reference fixture file ([] when there is none)."""
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "reference_*.txt"))):
        with open(path) as f:
            out += [r for r in (_parse_line(l) for l in f) if r]
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "reference_*.json"))):
        with open(path) as f:
            out += json.load(f)
    return out
