"""CPU oracle for the geoguessr-ai hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product package
(``geoguessr-ai_amd/``) never imports it and fails loudly when the HIP
library is missing.

Parity status (see DESIGN.md "Oracle"):

* ``geo_ref``   (haversine / smooth labels / soft-CE / head)  -- PINNED against
  the imported reference (``models/utils.py``, ``models/super_guessr.py``) via
  the fixtures under ``tests/golden`` (generator: ``tests/golden/make_golden.py``).
* ``clip_ref``  (CLIP vision tower)                            -- PINNED against
  ``transformers`` ``CLIPVisionModel`` (the library the reference calls,
  ``pretrain/clip_embedder.py:26,63-65``) on a tiny committed config.
* ``proto_ref`` (ProtoRefiner)                                 -- PINNED: helpers
  (``_euclidean_distance``, ``_temperature_softmax``, fp64 ``haversine``) and
  ``refine`` against the reference's own ``forward`` + ``_within_cluster_refinement``
  executed by ``tests/golden/make_golden_r4.py`` (``proto_refine.npz``).
* ``preprocess_ref`` (Pillow resampling + CLIP / timm / torchvision geometry)
                                                               -- PINNED against
  Pillow and ``transformers.CLIPImageProcessorPil`` (``preprocess_pil.npz``).
* ``tinyvit_ref`` (TinyViT)                                    -- PARITY UNPINNED:
  the arithmetic lives in ``timm==1.0.21`` (``uv.lock``), which is neither
  vendored in the reference nor installed here.  The restatement follows the
  published architecture and is self-checked by parameter
  counts, state-dict key table and MAC totals.  One component has an independent
  implementation in this image: timm's TinyViT ``Attention`` is LeViT's, and the
  attention core of this restatement equals ``transformers``' ``LevitAttention``
  (``tests/test_oracle_models.py::test_tinyvit_attention_core_matches_transformers_levit``).
"""
