"""Harness-side stand-in for the `soundfile` package (libsndfile binding, absent from this image) so that the REAL reference
dataset code (fairseq/data/audio/audio_utils.py:7-55) can run when data goldens are generated.  Implements the one call the
reference makes — read(path, dtype="float32", start=, frames=) on 16-bit PCM WAV — with the stdlib `wave` module and
libsndfile's documented int16 -> float32 scaling (sample / 32768).  Never imported by the product or by tests."""
import wave

import numpy as np


def read(file, dtype="float32", start=0, frames=-1, always_2d=False):
    assert dtype == "float32"
    with wave.open(file, "rb") as w:
        assert w.getsampwidth() == 2, "stub handles 16-bit PCM only"
        ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        start = min(max(start, 0), n)
        w.setpos(start)
        cnt = n - start if frames is None or frames < 0 else min(frames, n - start)
        raw = w.readframes(cnt)
    x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    if ch > 1 or always_2d:
        x = x.reshape(-1, ch)
    return x, sr
