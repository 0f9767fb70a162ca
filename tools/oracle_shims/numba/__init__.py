"""Import-only stub (test tooling)."""
class _Cuda:
    def jit(self, *a, **k):
        def deco(f): return f
        return deco if not (len(a) == 1 and callable(a[0])) else a[0]
    def __getattr__(self, name):
        raise RuntimeError("numba.cuda stub")
cuda = _Cuda()
