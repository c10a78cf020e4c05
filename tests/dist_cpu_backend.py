"""numpy stand-in for radix-sort_amd.dist.HipLocal -- TEST INFRASTRUCTURE: lets the splitter logic and
the exchange run on CPU tensors under gloo.  The product path never uses it."""
import numpy as np
import torch

from cpu_checkers import load_oracle

_U = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


class NumpyLocal:
    def __init__(self, key_bytes, transform):
        self.kb, self.xf, self.top = key_bytes, transform, key_bytes * 8
        self.oracle = load_oracle()

    def _order_bits(self, raw):
        u = raw.view(_U[self.kb])
        sign = np.array(1, dtype=u.dtype) << np.array(self.kb * 8 - 1, dtype=u.dtype)
        if self.xf == 1:
            return u ^ sign
        if self.xf == 2:
            return np.where((u & sign) != 0, ~u, u ^ sign)
        return u

    def _digit(self, raw, begin_bit, end_bit):
        return ((self._order_bits(raw) >> np.array(begin_bit, dtype=_U[self.kb])) &
                np.array((1 << (end_bit - begin_bit)) - 1, dtype=_U[self.kb])).astype(np.int64)

    def empty_like(self, t, n):
        return torch.empty(n, dtype=t.dtype)

    def histogram(self, t, lo, hi, begin_bit, end_bit):
        return np.bincount(self._digit(t.numpy()[lo:hi], begin_bit, end_bit), minlength=256).astype(np.int64)

    def partition(self, t, lo, hi, begin_bit, end_bit, scratch, vals=None, vscratch=None):
        a = t.numpy()
        order = np.argsort(self._digit(a[lo:hi], begin_bit, end_bit), kind="stable")
        a[lo:hi] = a[lo:hi][order]
        if vals is not None:
            v = vals.numpy()
            v[lo:hi] = v[lo:hi][order]

    def partition_out(self, t, out, begin_bit, end_bit, vals=None, vout=None):
        a = t.numpy()
        d = self._digit(a, begin_bit, end_bit)
        order = np.argsort(d, kind="stable")
        out.numpy()[:] = a[order]
        if vals is not None:
            vout.numpy()[:] = vals.numpy()[order]
        return np.bincount(d, minlength=256).astype(np.int64)

    def _typed(self, a):
        return a.view({4: np.float32, 8: np.float64}[self.kb]) if self.xf == 2 else a.view(_U[self.kb])

    def sort(self, t, scratch=None, vals=None, vscratch=None):
        a = t.numpy()
        if vals is None:
            a[:] = self.oracle.sort_keys(self._typed(a), self.xf, mt=True).view(a.dtype)
        else:
            k, perm = self.oracle.sort_pairs(self._typed(a), self.xf, mt=True)     # stable; perm = source positions
            v = vals.numpy()
            v[:] = v[perm.astype(np.int64)]
            a[:] = k.view(a.dtype)

    def single_value(self, t, lo, hi, below_bit):
        if hi <= lo:
            return 0, 0
        if below_bit <= 0:
            return 1, 0
        low = self._order_bits(t.numpy()[lo:hi]).astype(np.uint64) & np.uint64((1 << below_bit) - 1)
        return (1, int(low[0])) if bool((low == low[0]).all()) else (2, 0)

    def synchronize(self):
        pass

    def wait_for_producers(self, t):
        pass
