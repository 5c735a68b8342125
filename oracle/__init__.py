"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference hot path (see rsqc_oracle.c)."""
