"""Environment-level pieces of the hot path (term library + step harness)."""
