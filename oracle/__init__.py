"""CPU oracle package (test infrastructure only -- never imported by acme_jl_amd)."""
