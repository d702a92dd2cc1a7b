"""Mirror of the reference package path supernnova.training (model classes)."""
