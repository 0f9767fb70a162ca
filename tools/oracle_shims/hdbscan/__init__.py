"""Import-only stub (test tooling)."""
class HDBSCAN:
    def __init__(self, *a, **k):
        raise RuntimeError("hdbscan stub")
