"""GPU parity against the committed golden fixtures (outputs of the reference itself)."""
import numpy as np
import pytest

import golden_cases as g
from radix_sort_amd import shapes

pytestmark = pytest.mark.gpu


def test_golden_keys(engine):
    for e in g.MANIFEST["keys"] + g.MANIFEST["edge_keys"]:
        v = g.key_input(e)
        out = engine.sort_keys_host(v, shapes.KEY_TYPES[e["type"]][2])
        assert g.sha(out) == e["expected_sha256"], e


def test_golden_pairs(engine):
    for e in g.MANIFEST["pairs"]:
        v = g.key_input(e)
        k, perm = engine.sort_pairs_host(v, transform=shapes.KEY_TYPES[e["type"]][2])
        assert g.sha(k) == e["keys_sha256"] and g.sha(perm) == e["perm_sha256"], e
