"""vietTTS/hifigan/config.py:3-6 — FLAGS.ckpt_dir of the vocoder checkpoint."""
from viettts_amd.hifigan.config import FLAGS  # noqa: F401
