"""Mirrors of friture/signal/*.py on the HIP backend (same function names and argument meaning)."""
