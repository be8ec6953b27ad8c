"""The recorded profile counters (profiles/pmc_latest.json: what bench.py replays as roofline.traffic /
clock / cache figures) carry the hashes of the kernel sources they were captured from; bench.py marks
them stale when the tree differs ("nothing enforces that").  Here: the hashing is
deterministic and sensitive, the committed record has its hashes, and a mismatch is reported loudly."""
import json
import os
import shutil
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unit_hash_covers_the_unit_and_its_headers(tmp_path, monkeypatch):
    from schnorr_amd import build as B
    a = B.unit_sources_sha256("k_verify.hip")
    assert a == B.unit_sources_sha256("k_verify.hip") and len(a) == 64
    assert a != B.unit_sources_sha256("k_hash.hip")
    # a copy of the source tree with one byte changed in a header k_verify.hip includes, and in one it
    # does not include
    csrc = tmp_path / "csrc"
    shutil.copytree(B.CSRC, csrc, ignore=shutil.ignore_patterns("__pycache__"))
    os.makedirs(tmp_path / "include", exist_ok=True)
    monkeypatch.setattr(B, "CSRC", str(csrc))
    base = B.unit_sources_sha256("k_verify.hip")
    with open(csrc / "fe29.h", "a") as f:
        f.write("\n// touched\n")
    assert B.unit_sources_sha256("k_verify.hip") != base
    base = B.unit_sources_sha256("k_verify.hip")
    with open(csrc / "inv29.h", "a") as f:     # only k_misc.hip includes it
        f.write("\n// touched\n")
    assert B.unit_sources_sha256("k_verify.hip") == base


def test_committed_counters_name_the_build_they_are_valid_for():
    from schnorr_amd import build as B
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
        rec = json.load(f)
    ev = rec.get("evidence")
    assert ev and len(ev.get("k_verify_sources_sha256", "")) == 64 and len(ev.get("k_hash_sources_sha256", "")) == 64
    assert rec.get("commit") and rec.get("captured")
    now = B.evidence_hashes()
    if ev["k_verify_sources_sha256"] != now["k_verify_sources_sha256"]:
        # not a failure of the code: the counters must be re-captured (tools/gpu_round_check.sh) — bench.py
        # reports roofline.pmc_source.stale = true until then
        warnings.warn("profiles/pmc_latest.json was captured from other kernel sources (commit %s): "
                      "bench.py will mark the replayed counters stale" % rec.get("commit"))
