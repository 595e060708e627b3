"""Side-channel state carried through every forward -- same fields as the reference's
``StreamingArgs`` (src/streaming_vlm/inference/streaming_args.py:1-10)."""
from __future__ import annotations


class StreamingArgs:
    def __init__(self, pos_mode: str, all_text: bool = False):
        if pos_mode not in ("append", "shrink"):
            raise AssertionError("pos_mode must be in ['append', 'shrink']")
        self.pos_mode = pos_mode          # "shrink": positions stay contiguous after eviction; "append": grow forever
        self.all_text = all_text
        self.input_ids = None             # full (pruned) ids of the stream; shrink mode derives positions from them
        self.video_grid_thw = None        # cumulative, one row per chunk (never pruned, inference.py:415)
        self.second_per_grid_ts = None
        self.last_cache_position = -1
