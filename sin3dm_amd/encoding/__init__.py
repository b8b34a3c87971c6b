"""Autoencoder decode side of the Sin3DM path on MI355X (mirrors the reference package src/encoding)."""
