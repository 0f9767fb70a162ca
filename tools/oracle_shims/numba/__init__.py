"""Stand-in for numba (absent from the image), test tooling only.  `cuda.jit` kernels are EMULATED on the CPU: the launcher
`kernel[blocks, threads](*args)` runs the reference's own kernel body once per thread of the grid, with `cuda.grid(2)`
returning that thread's coordinates -- so goldens for the numba-CUDA kernels of src/graph_balancer/ricci.py come from the
reference's code itself, not from a restatement.

Typing: numba promotes `int literal (int64) <op> float32` to float64 and rounds when the result is stored into a float32
array.  Array elements are therefore handed to the kernel body as Python floats (float64) and rounded on store, which
reproduces that promotion for these kernels (their float32 x float32 products are products of small integers: exact)."""
import itertools
import threading

_tls = threading.local()


class _Arr:
    """ndarray view whose element reads are Python scalars (float64 / int), stores round to the array dtype."""
    def __init__(self, a):
        self.a = a

    def __getitem__(self, idx):
        return self.a[idx].item()

    def __setitem__(self, idx, v):
        self.a[idx] = v

    @property
    def shape(self):
        return self.a.shape


class _Launcher:
    def __init__(self, fn):
        self.fn = fn

    def __getitem__(self, cfg):
        blocks, threads = cfg

        def launch(*args):
            args = [a.numpy() if hasattr(a, 'numpy') and hasattr(a, 'detach') else a for a in args]
            args = [a.item() if hasattr(a, 'item') and getattr(a, 'ndim', 1) == 0 else a for a in args]
            args = [_Arr(a) if hasattr(a, 'ndim') and getattr(a, 'ndim', 0) >= 1 else a for a in args]
            for bx, by in itertools.product(range(blocks[0]), range(blocks[1])):
                for tx, ty in itertools.product(range(threads[0]), range(threads[1])):
                    _tls.pos = (bx * threads[0] + tx, by * threads[1] + ty)
                    self.fn(*args)
        return launch

    def __call__(self, *a, **k):
        return self.fn(*a, **k)


class _Cuda:
    def jit(self, *a, **k):
        if len(a) == 1 and callable(a[0]):
            return _Launcher(a[0])
        return lambda f: _Launcher(f)

    def grid(self, ndim):
        return _tls.pos if ndim == 2 else _tls.pos[0]


cuda = _Cuda()
