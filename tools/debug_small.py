import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import radix_sort_amd as rs
from radix_sort_amd import shapes
from cpu_checkers import load_oracle
orc = load_oracle(); eng = rs.Engine(0)
bad = 0
for name in ("u32", "u8", "u16", "u64", "i32", "f32"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    for n in (2, 10, 100, 1000, 8192, 8193, 9999, 16384, 16385, 20000, 100000, 300000, 2100000, 2200000):
        v = shapes.generate(name, n, "randomized")
        want = orc.sort_keys(v, xf, mt=True)
        try:
            got = eng.sort_keys_host(v, xf)
        except Exception as e:
            print(name, n, "EXC", e); bad += 1; continue
        if got.tobytes() != want.tobytes():
            bad += 1
            diff = np.nonzero(got.view(np.uint8).reshape(n, kb) != want.view(np.uint8).reshape(n, kb))[0]
            print(name, n, "MISMATCH first idx", diff[0], "count", len(np.unique(diff)), "sorted?", bool(np.all(got[:-1] <= got[1:])) if xf == 0 else "-",
                  "multiset ok?", np.array_equal(np.sort(got.view(np.uint8 if kb==1 else {2:np.uint16,4:np.uint32,8:np.uint64}[kb])), np.sort(want.view({1:np.uint8,2:np.uint16,4:np.uint32,8:np.uint64}[kb]))))
        else:
            print(name, n, "ok")
        if bad >= 6: sys.exit(1)
print("bad", bad)
