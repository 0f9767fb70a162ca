"""Import-only stub (test tooling)."""
class Data:
    def __init__(self, *a, **k):
        raise RuntimeError("torch_geometric stub")
