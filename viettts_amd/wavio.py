"""Minimal RIFF/WAVE writer — what ``soundfile.write(path, wave, samplerate)`` does for a float array
and a ``.wav`` path in the reference CLI (vietTTS/synthesizer.py:39): libsndfile's default subtype for
WAV is PCM_16, mono for a 1-D array (the shipped assets/infore/clip.wav is 16 kHz PCM16 mono).
``soundfile`` is not installable offline, hence this 30-line stand-in."""
from __future__ import annotations

import struct

import numpy as np


def float_to_pcm16(wave: np.ndarray) -> np.ndarray:
    """libsndfile float -> PCM16 with its default normalisation: scale by 0x8000... (clip on),
    i.e. round(x * 32767) clipped to int16."""
    x = np.asarray(wave, dtype=np.float64)
    y = np.rint(np.clip(x, -1.0, 1.0) * 32767.0)
    return y.astype("<i2")


def write_wav(path, wave: np.ndarray, samplerate: int) -> None:
    wave = np.asarray(wave)
    if wave.ndim != 1:
        raise ValueError("write_wav expects a mono 1-D waveform")
    pcm = float_to_pcm16(wave).tobytes()
    n = len(pcm)
    hdr = b"RIFF" + struct.pack("<I", 36 + n) + b"WAVE"
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, int(samplerate), int(samplerate) * 2, 2, 16)
    hdr += b"data" + struct.pack("<I", n)
    with open(str(path), "wb") as f:
        f.write(hdr + pcm)


def read_wav(path):
    """(samplerate, int16 array) — for tests."""
    with open(str(path), "rb") as f:
        b = f.read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    fmt, ch, sr, _, _, bits = struct.unpack("<HHIIHH", b[20:36])
    assert fmt == 1 and ch == 1 and bits == 16
    n = struct.unpack("<I", b[40:44])[0]
    return sr, np.frombuffer(b[44 : 44 + n], dtype="<i2")
