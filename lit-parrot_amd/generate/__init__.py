"""Decode loop (reference: generate/base.py)."""
