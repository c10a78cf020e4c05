"""Per-pass device times of one device-resident keys-only sort: python tools/pass_times.py <type> <shape> [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sort_amd as rs
from radix_sort_amd import shapes

name, shape = sys.argv[1], sys.argv[2]
n = 1 << int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
begin_bit = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # sort on the bits [begin_bit, end_bit) only
end_bit = int(sys.argv[5]) if len(sys.argv) > 5 else None
dt, kb, xf = shapes.KEY_TYPES[name]
eng = rs.Engine(0)
eng.set_precheck(False)
eng.set_profiling(True)
src = shapes.generate_device(name, n, shape, "cuda")
work = torch.empty_like(src)
for it in range(3):
    work.copy_(src); torch.cuda.synchronize()
    eng.sort_keys_ptr(work.data_ptr(), n, kb, xf, begin_bit=begin_bit, end_bit=end_bit); eng.synchronize()
    p = eng.profile()
print(name, shape, n, "hist %.3f" % p[0], "passes", " ".join("%.3f" % x for x in p[1:]), "total %.3f" % sum(p))
