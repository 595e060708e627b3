"""WebVTT helpers with the reference's output format (src/streaming_vlm/utils/vtt_utils.py:5-16)."""
from __future__ import annotations

import os
from contextlib import contextmanager


@contextmanager
def open_vtt(path):
    """Append to `path`, writing the WEBVTT header first when the file is new."""
    new = not os.path.exists(path)
    with open(path, "w" if new else "a", encoding="utf-8") as f:
        if new:
            f.write("WEBVTT\n\n")
        yield f


def sec2ts(sec: float) -> str:
    whole = int(sec)
    ms = int((sec - whole) * 1000)
    return f"{whole // 3600:02d}:{whole % 3600 // 60:02d}:{whole % 60:02d}.{ms:03d}"
