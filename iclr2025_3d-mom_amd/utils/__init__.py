"""Host-side helpers mirroring the reference's `utils` package (only what the 4DGS hot path uses)."""
