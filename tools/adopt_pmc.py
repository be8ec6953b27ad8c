#!/usr/bin/env python3
"""Adopt a PMC summary recorded on the GPU box (tools/gpu_round_check.sh writes gpurun_out/<round>_pmc_latest.json;
the box has no .git) as profiles/pmc_latest.json: checks that its source hashes are those of THIS tree and
fills in the commit the tree is at.

    python tools/adopt_pmc.py gpurun_out/r06_pmc_latest.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from schnorr_amd import build as B  # noqa: E402

rec = json.load(open(sys.argv[1]))
here = B.evidence_hashes()
ev = rec.get("evidence", {})
for k in ("k_verify_sources_sha256", "k_hash_sources_sha256"):
    if ev.get(k) != here[k]:
        sys.exit("%s: recorded %s, this tree %s — the counters are not valid for this build" % (k, ev.get(k), here[k]))
rec["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, text=True).strip()
with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as f:
    json.dump(rec, f, indent=1)
    f.write("\n")
print("profiles/pmc_latest.json: captured %s, commit %s" % (rec.get("captured"), rec["commit"]))
