"""Which sweep kernels does a small sort run when RSORT_CHAIN_MIN_TILES lowers the chain-mode bound?  Run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sort_amd as rs
from radix_sort_amd import shapes
eng = rs.Engine(0); eng.set_precheck(False)
for name in ("u32", "u64"):
    dt, kb, xf = shapes.KEY_TYPES[name]
    v = shapes.generate(name, 300001, "randomized")
    eng.sort_keys_host(v, xf)
