"""Token-pattern span scanner (oracle; test infrastructure only).

Restates ``src/streaming_vlm/utils/get_qwen_range.py:15-86``: a left-to-right
scan that returns the k-th CLOSED span ``[start, end]`` of a chat-template
segment.  Token ids are the hard-coded Qwen ids of ``get_qwen_range.py:2-13``.
"""
from __future__ import annotations

SYSTEM_PROMPT_OFFSET = 58          # get_qwen_range.py:1
IM_START = 151644
IM_END = 151645
USER = 872
ASSISTANT = 77091
VISION_START = 151652
VISION_END = 151653
VIDEO_PAD = 151656
LF = 198
PREVIOUS_TEXT = (19702, 1467)
TIME = 1462

_PATTERNS = {
    # label: (start pattern, end pattern)            get_qwen_range.py:38-63
    "user": ((IM_START, USER), (IM_END,)),
    "previous text": ((IM_START,) + PREVIOUS_TEXT + (LF,), (IM_END,)),
    "user_text": ((TIME,), (VISION_START,)),
    "assistant": ((IM_START, ASSISTANT), (IM_END,)),
    "vision": ((VISION_START,), (VISION_END,)),
}


def all_ranges(ids, label: str, contain_lf: bool = True):
    """All spans of `label`, in order of appearance (closed intervals)."""
    assert label in _PATTERNS
    ids = [int(t) for t in ids]
    sp, ep = _PATTERNS[label]
    ls, le = len(sp), len(ep)
    n = len(ids)
    segs = []
    i = 0
    while i <= n - ls:                                   # :69
        if tuple(ids[i:i + ls]) == sp:
            j = i + ls
            found = False
            while j <= n - le:                           # :74
                if tuple(ids[j:j + le]) == ep:
                    # one trailing "\n" belongs to the span when present and wanted  (:76-79)
                    if j + le < n and ids[j + le] == LF and contain_lf:
                        segs.append((i, j + le))
                    else:
                        segs.append((i, j + le - 1))
                    i = j + le                           # :80 (the "\n" itself is re-scanned, harmlessly)
                    found = True
                    break
                j += 1
            if not found:                                # while/else: unterminated segment ends the scan (:82-83)
                break
        else:
            i += 1
    if label == "user_text":                             # :84-85 -- text ends before <|vision_start|>
        segs = [(s, e - 1) for (s, e) in segs]
    return segs


def get_qwen_range(ids, label: str, index: int, contain_lf: bool = True):
    """k-th span (negative k allowed); raises IndexError like the reference."""
    return all_ranges(ids, label, contain_lf)[index]
