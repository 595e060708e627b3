"""CPU oracle for the streaming-VLM hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain eager-PyTorch / numpy restatement of the algorithm of
rahim-xelpmoc/streaming-vlm's per-chunk streaming loop (reference files are
cited function by function as ``file:line`` relative to ``/root/reference``).

Rules (enforced by tests/test_abi_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * the product package (``streaming-vlm_amd/``) never imports it and never
    falls back to it: the HIP path fails loudly when its extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``qwen_range`` / ``rope_index`` / ``vtt`` are pinned against the reference's
    own importable modules (golden vectors in ``tests/golden``; generator
    ``oracle/make_golden.py``).
  * the model math (ViT, decoder, M-RoPE, RMSNorm ...) lives in third-party
    ``transformers==4.52.4`` + ``flash_attn 2.8`` which are NOT part of the
    reference tree; it is pinned against the stock ``transformers`` modules
    installed in the build container on a tiny seeded config (golden vectors
    committed), and by the shrink-mode invariant
    (post-cache RoPE on un-rotated K == pre-cache RoPE from scratch).
  * the reference ships no tests or golden vectors of its own, so by the
    reference's own fixtures parity is unpinned; everything above is pinned by
    vectors minted from the importable pieces.
"""
