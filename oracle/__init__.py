"""Test infrastructure only: CPU restatement of the reference's hot path (see bert4rec_oracle.py)."""
