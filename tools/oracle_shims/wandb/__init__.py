"""Import-only stub (test tooling): lets the reference's modules import; logging calls are no-ops."""
class _Run:
    def __getattr__(self, name):
        return lambda *a, **k: None
def init(*a, **k): return _Run()
def log(*a, **k): return None
def finish(*a, **k): return None
class Table:
    def __init__(self, *a, **k): pass
class Histogram:
    def __init__(self, *a, **k): pass
class Video:
    def __init__(self, *a, **k): pass
class Image:
    def __init__(self, *a, **k): pass
run = None
