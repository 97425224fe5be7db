"""vietTTS/nat/config.py:8-45 — FLAGS (the fields the inference path reads)."""
from viettts_amd.nat.config import FLAGS  # noqa: F401
