"""vietTTS/hifigan/mel2wave.py:20 — same module path, name and positional signature; the generator runs in the HIP library."""
from viettts_amd.hifigan.mel2wave import mel2wave, reload  # noqa: F401
