"""vietTTS/nat/text2mel.py:16-102 — predict_duration, text2tokens, predict_mel, text2mel with the reference's signatures;
the two networks run in the HIP library (include/vtts_nat.h)."""
from viettts_amd.nat.text2mel import load_lexicon, predict_duration, predict_mel, text2mel, text2tokens  # noqa: F401
