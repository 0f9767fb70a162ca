"""Import-only stub (test tooling)."""
class TFRecordDataset:
    def __init__(self, *a, **k):
        raise RuntimeError("tfrecord stub: datasets are not available in this image")
