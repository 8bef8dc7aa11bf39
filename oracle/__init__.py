"""CPU oracle (test infrastructure only; see oracle/vit_ref.py header)."""
