"""Mirror of the reference package path supernnova.validation (hot-path subset)."""
