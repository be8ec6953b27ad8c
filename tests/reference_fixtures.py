"""Parser for fixtures produced by the REAL dusk-schnorr 0.18 — TEST INFRASTRUCTURE.

Nothing under /root/reference pins the challenge hash; the reference cannot be
built in the authoring image.  A maintainer with cargo runs
    cargo run --features golden --bin golden_gen > tests/golden/reference_vectors.txt   (rust/dusk-schnorr-gpu)
and drops the file in; from then on tests/test_oracle.py::test_reference_fixtures_pin_the_oracle
and tests/test_gpu_parity.py::test_reference_fixtures_on_gpu compare the oracle and the HIP engine
with it and parity is pinned without any code change.  Any file tests/golden/reference_*.txt in
golden_gen.rs's line format is picked up; tests/golden/reference_*.json (a list of the same
records as dicts) as well.

Line format (one record per line, fields separated by blanks, hex = little-endian to_bytes(),
points as affine u || v):
    sponge_hash n <k> le <hex32>            dusk_poseidon::sponge::hash(&[1, .., k])
    truncated_hash n <k> le <hex32>         sponge::truncated::hash of the same inputs
    sponge_hash_1_2_3_le <hex32>            (r03 names of the k = 3 records)
    truncated_hash_1_2_3_le <hex32>
    sig  <i> sk <32> m <32> u <32> R <64> PK <64> [c <32>] sig_bytes <64> pk_bytes <32> verdict <true|false>
    sigd <i> sk <32> m <32> u <32> R <64> Rp <64> PK <64> PKp <64> c <32> sig_bytes <96> pk_bytes <64> verdict <..>
    sigv <i> sk_bytes <64> m <32> u <32> R <64> PK <64> Gen <64> c <32> sig_bytes <64> pk_bytes <64> verdict <..>
    stdrng <seed> first <n> <hex n>         the first n bytes of StdRng::seed_from_u64(seed)
    wide fr|fq <hex64> <hex32>              Field::random over an rng returning exactly those 64 bytes
    from_bytes <i> <hex32> ok u <hex32> v <hex32>
    from_bytes <i> <hex32> err
"""
import glob
import json
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = ("sponge_hash", "truncated_hash", "sig", "sigd", "sigv", "stdrng", "wide", "from_bytes")


def _parse_line(line):
    t = line.split()
    if not t:
        return None
    if t[0] in ("sponge_hash_1_2_3_le", "truncated_hash_1_2_3_le"):
        return {"kind": t[0].split("_1_2_3")[0], "n": 3, "hex": t[1]}
    if t[0] in ("sponge_hash", "truncated_hash"):
        if t[1] != "n" or t[3] != "le":
            raise ValueError("malformed %s line" % t[0])
        return {"kind": t[0], "n": int(t[2]), "hex": t[4]}
    if t[0] in ("sig", "sigd", "sigv"):
        rec = {"kind": t[0], "i": int(t[1])}
        for k, v in zip(t[2::2], t[3::2]):
            rec[k] = v
        rec["verdict"] = rec["verdict"] == "true"
        return rec
    if t[0] == "stdrng":
        if t[2] != "first" or len(t[4]) != 2 * int(t[3]):
            raise ValueError("malformed stdrng line")
        return {"kind": "stdrng", "seed": int(t[1]), "n": int(t[3]), "hex": t[4]}
    if t[0] == "wide":
        if t[1] not in ("fr", "fq") or len(t[2]) != 128 or len(t[3]) != 64:
            raise ValueError("malformed wide line")
        return {"kind": "wide", "field": t[1], "wide": t[2], "hex": t[3]}
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


def format_record(r):
    """the golden_gen.rs line of a record dict (used to build synthetic files from predictions)"""
    k = r["kind"]
    if k in ("sponge_hash", "truncated_hash"):
        return "%s n %d le %s" % (k, r["n"], r["hex"])
    if k == "sig":
        return ("sig %d sk %s m %s u %s R %s PK %s c %s sig_bytes %s pk_bytes %s verdict %s"
                % (r["i"], r["sk"], r["m"], r["u"], r["R"], r["PK"], r["c"], r["sig_bytes"], r["pk_bytes"],
                   "true" if r["verdict"] else "false"))
    if k == "sigd":
        return ("sigd %d sk %s m %s u %s R %s Rp %s PK %s PKp %s c %s sig_bytes %s pk_bytes %s verdict %s"
                % (r["i"], r["sk"], r["m"], r["u"], r["R"], r["Rp"], r["PK"], r["PKp"], r["c"], r["sig_bytes"],
                   r["pk_bytes"], "true" if r["verdict"] else "false"))
    if k == "sigv":
        return ("sigv %d sk_bytes %s m %s u %s R %s PK %s Gen %s c %s sig_bytes %s pk_bytes %s verdict %s"
                % (r["i"], r["sk_bytes"], r["m"], r["u"], r["R"], r["PK"], r["Gen"], r["c"], r["sig_bytes"],
                   r["pk_bytes"], "true" if r["verdict"] else "false"))
    if k == "stdrng":
        return "stdrng %d first %d %s" % (r["seed"], r["n"], r["hex"])
    if k == "wide":
        return "wide %s %s %s" % (r["field"], r["wide"], r["hex"])
    if k == "from_bytes":
        return ("from_bytes %d %s ok u %s v %s" % (r["i"], r["enc"], r["u"], r["v"]) if r["ok"]
                else "from_bytes %d %s err" % (r["i"], r["enc"]))
    raise ValueError(k)
