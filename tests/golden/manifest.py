"""tests/golden/manifest.py -- content hashes of the committed fixtures (round 6, VERDICT r5 item 2: two fixtures had gone stale against their own generators and no test noticed).

    python tests/golden/manifest.py write      # after regenerating fixtures: rewrites tests/golden/MANIFEST.json
    python tests/golden/manifest.py check      # exit status 1 if a fixture's content differs from the manifest

A fixture's hash is taken over its CONTENT -- sorted (key, dtype, shape, bytes) of every array -- not over the .npz file: the zip container stamps every member with the
time of writing, so two bit-identical regenerations give different files.  The manifest also records, per fixture, the generator script that makes it and the git blob
hashes of the sources the oracle-produced fields depend on (oracle/lmpc_oracle.py, oracle/osqp_restated.c): tests/test_oracle_golden.py::test_manifest_* compares the
hashes, checks that the oracle sources are the ones the fixtures were made with, and re-derives the oracle-produced fields (`sol_opt`, `y_opt`, `cert_opt`) of a sample of
records bit for bit.  Rule: an edit of oracle/ that changes osqp_solve_exact re-makes the fixtures (and this manifest) in the same commit.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PATH = os.path.join(HERE, "MANIFEST.json")
GENERATORS = {
    "lmpc_n12.npz": "make_golden.py", "ltvmpc_n12.npz": "make_golden.py",
    "lmpc_wide_n12.npz": "make_wide_golden.py", "lmpc_n14.npz": "make_wide_golden.py", "lmpc_n40.npz": "make_wide_golden.py", "lmpc_30laps_n12.npz": "make_wide_golden.py",
    "lmpc_30laps_stress_n12.npz": "make_wide_golden.py", "mpc_n14.npz": "make_wide_golden.py",
    "ltvmpc_noslack_n12.npz": "make_noslack_golden.py", "regression.npz": "make_regression_golden.py", "track_xy.npz": "make_track_golden.py",
    "reference_flow_laps_n14.json": "make_flow_golden.py",
    "reg_singular_capture.npz": "tools/capture_reg_singular.py (GPU box: captured closed-loop events, not regenerable here)",
}
ORACLE_SOURCES = ("oracle/lmpc_oracle.py", "oracle/osqp_restated.c")


def content_hash(path):
    h = hashlib.sha256()
    if path.endswith(".npz"):
        with np.load(path) as g:
            for k in sorted(g.files):
                a = np.ascontiguousarray(g[k])
                h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(repr(a.shape).encode()); h.update(a.tobytes())
    else:
        h.update(open(path, "rb").read())
    return h.hexdigest()


def git_blob_hash(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def current():
    fixtures = {}
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz") or f == "reference_flow_laps_n14.json":
            fixtures[f] = {"content_sha256": content_hash(os.path.join(HERE, f)), "generator": GENERATORS.get(f, "?")}
    return {"fixtures": fixtures, "oracle_sources": {p: git_blob_hash(os.path.join(ROOT, p)) for p in ORACLE_SOURCES},
            "generators": {f: git_blob_hash(os.path.join(HERE, f)) for f in sorted(set(v for v in GENERATORS.values() if v.endswith(".py") and "/" not in v))}}


def load():
    with open(PATH) as f:
        return json.load(f)


def diff():
    """[(what, name)] for everything that differs between the tree and MANIFEST.json."""
    m, c = load(), current()
    out = []
    for sec in ("fixtures", "oracle_sources", "generators"):
        for k in sorted(set(m[sec]) | set(c[sec])):
            if m[sec].get(k) != c[sec].get(k):
                out.append((sec, k))
    return out


if __name__ == "__main__":
    if sys.argv[1:] == ["write"]:
        with open(PATH, "w") as f:
            json.dump(current(), f, indent=1, sort_keys=True)
        print("wrote", PATH)
    else:
        d = diff()
        for sec, k in d:
            print("differs from MANIFEST.json: %s %s" % (sec, k))
        sys.exit(1 if d else 0)
