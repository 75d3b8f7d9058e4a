"""CPU-only studies built on the oracle (test infrastructure, never imported by the product)."""
