"""vietTTS/hifigan/convert_torch_model_to_haiku.py:65-79 — same module path and flags (scripts/quick_start.sh:7)."""
from viettts_amd.hifigan.convert_torch_model_to_haiku import convert_to_haiku, load_checkpoint, main  # noqa: F401

if __name__ == "__main__":
    main()
