"""vietTTS/synthesizer.py:12-39 — ``python -m vietTTS.synthesizer --text STR --output PATH --sample-rate INT
--silence-duration FLOAT --lexicon-file PATH``: same flags, defaults and prints."""
from viettts_amd.synthesizer import build_parser, main, nat_normalize_text  # noqa: F401

if __name__ == "__main__":
    raise SystemExit(main())
