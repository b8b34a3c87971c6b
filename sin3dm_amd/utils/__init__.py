"""Host-side helpers mirroring the reference package src/utils."""
