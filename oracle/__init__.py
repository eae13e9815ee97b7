"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see graph_oracle.c header)."""
