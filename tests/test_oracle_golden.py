"""CPU: the oracle (oracle/rs_oracle.cpp) reproduces every golden vector the reference build made."""
import numpy as np
import pytest

import golden_cases as g
from radix_sort_amd import shapes


def _ids(entries):
    return [f"{e.get('type', 'str')}-{e.get('shape', e.get('label', ''))}-{e['n']}" for e in entries]


@pytest.mark.parametrize("e", g.MANIFEST["keys"] + g.MANIFEST["edge_keys"], ids=_ids(g.MANIFEST["keys"] + g.MANIFEST["edge_keys"]))
def test_keys(oracle, e):
    v = g.key_input(e)
    xf = shapes.KEY_TYPES[e["type"]][2]
    for mt in (False, True):
        out = oracle.sort_keys(v, xf, mt=mt)
        assert g.sha(out) == e["expected_sha256"]
        if "npz" in e:
            assert out.tobytes() == g.small()[e["npz"] + "/out"].tobytes()


@pytest.mark.parametrize("e", g.MANIFEST["pairs"], ids=_ids(g.MANIFEST["pairs"]))
def test_pairs(oracle, e):
    v = g.key_input(e)
    xf = shapes.KEY_TYPES[e["type"]][2]
    k, perm = oracle.sort_pairs(v, xf, mt=True)
    assert g.sha(k) == e["keys_sha256"] and g.sha(perm) == e["perm_sha256"]
    assert (v[perm.astype(np.int64)].tobytes() == k.tobytes())


@pytest.mark.parametrize("e", g.MANIFEST["strings"] + g.MANIFEST["edge_strings"],
                         ids=_ids(g.MANIFEST["strings"] + g.MANIFEST["edge_strings"]))
def test_strings(oracle, e):
    blob, off = g.string_input(e)
    p0 = oracle.sort_strings(blob, off, stable=False)
    assert g.sorted_strings_sha(blob, off, p0) == e["sorted_sha256"]
    p1 = oracle.sort_strings(blob, off, stable=True)
    assert g.sha(p1) == e["perm_sha256"]
