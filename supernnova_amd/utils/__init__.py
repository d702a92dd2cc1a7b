"""Mirror of the reference package path supernnova.utils (hot-path subset)."""
