"""Access to tests/golden/ (fixtures produced by the reference build; see golden/make_golden.py)."""
import hashlib
import json
import os

import numpy as np

from radix_sort_amd import shapes

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(HERE, "manifest.json")) as _f:
    MANIFEST = json.load(_f)
_small = None


def small():
    global _small
    if _small is None:
        _small = np.load(os.path.join(HERE, "golden_small.npz"), allow_pickle=False)
    return _small


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def key_input(entry):
    """The input array of a manifest entry: stored bytes for small cases, else regenerated from
    (type, shape, n, seed) and verified against the recorded SHA-256."""
    if "npz" in entry:
        v = small()[entry["npz"] + "/in"]
    else:
        v = shapes.generate(entry["type"], entry["n"], entry["shape"], entry["seed"])
    assert sha(v) == entry["input_sha256"], "golden input does not reproduce"
    return v


def string_input(entry):
    if "npz" in entry:
        blob = small()[entry["npz"] + "/blob"]
        off = small()[entry["npz"] + "/off"].astype(np.uint64)
    else:
        blob, off = shapes.generate_strings(entry["n"], entry["shape"], seed=entry["seed"])
    if "input_sha256" in entry:
        assert sha(blob) + sha(off) == entry["input_sha256"], "golden input does not reproduce"
    return blob, off


def sorted_strings_sha(blob, off, perm):
    strs = shapes.unpack_strings(blob, off)
    ob, oo = shapes.pack_strings([strs[i] for i in perm.tolist()])
    return sha(ob) + sha(oo)
