"""acme_jl_amd -- MI355X-native batched ``run!`` for ACME.jl circuit models."""
