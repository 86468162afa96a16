"""Mirror of the reference package `heal_swin.models_torch` (hot-path modules only)."""
