"""Host-side mirror of the reference's `epoch_loops` package for the BMHRL mode (per-batch steps and decoders)."""
