"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/gmr_oracle.c header)."""
