"""TEST INFRASTRUCTURE ONLY: CPU restatement of the UTree SEARCH_GG path (see utree_oracle.h)."""
