"""CPU oracle (test infrastructure only) -- see ref_ops.py and afd_oracle.c headers."""
