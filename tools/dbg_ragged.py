"""Developer aid: where does a keys-only sort differ from numpy's?  usage: python tools/dbg_ragged.py <dtype> <n> [shape]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sort_amd as rs
from radix_sort_amd import shapes

def main():
    name = sys.argv[1]; shape = sys.argv[3] if len(sys.argv) > 3 else "randomized"
    eng = rs.Engine(0)
    for n in [int(x) for x in sys.argv[2].split(",")]:
        dt, kb, xf = shapes.KEY_TYPES[name]
        v = shapes.generate(name, n, shape)
        got = eng.sort_keys_host(v, xf)
        want = np.sort(v, kind="stable")
        bad = np.nonzero(got != want)[0]
        print(name, shape, n, "mismatches", bad.size, "first", bad[:8].tolist(), "last", bad[-4:].tolist())
        if bad.size:
            gu, gc = np.unique(got, return_counts=True); wu, wc = np.unique(want, return_counts=True)
            gd = dict(zip(gu.tolist(), gc.tolist())); wd = dict(zip(wu.tolist(), wc.tolist()))
            diff = {k: gd.get(k, 0) - wd.get(k, 0) for k in set(gd) | set(wd) if gd.get(k, 0) != wd.get(k, 0)}
            print("  multiset diff (got - want), up to 12:", sorted(diff.items())[:12], "total distinct differing", len(diff))
            print("  ordered?", bool((np.diff(got.astype(np.int64)) >= 0).all()) if got.dtype.kind in "ui" else "n/a")
    eng.close()
main()
