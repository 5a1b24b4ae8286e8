"""CPU oracle (test infrastructure only; see inr_oracle.py header)."""
