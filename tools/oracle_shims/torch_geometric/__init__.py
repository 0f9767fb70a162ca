"""Import-only stub (test tooling)."""
