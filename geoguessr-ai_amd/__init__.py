"""MI355X-native (gfx950) hot path of CogitoNTNU/geoguessr-ai: TinyViT / CLIP-ViT encoder forward+backward over
4-heading street-view batches feeding the SuperGuessr geocell classifier and ProtoRefiner, as hand-written HIP
kernels behind a C-ABI (``include/gg.h``, ``lib/libgg.so``) with the reference's Python call surface:

    geoguessr_ai_amd.models.tinyvit.TinyViTAdapter          <-> models/tinyvit.py
    geoguessr_ai_amd.models.super_guessr.SuperGuessr        <-> models/super_guessr.py
    geoguessr_ai_amd.models.proto_refiner.ProtoRefiner      <-> models/proto_refiner.py
    geoguessr_ai_amd.models.utils                           <-> models/utils.py
    geoguessr_ai_amd.pretrain.clip_embedder.CLIPEmbedding   <-> pretrain/clip_embedder.py
    geoguessr_ai_amd.pretrain.tinyvit_embedder              <-> pretrain/tinyvit_embedder.py
    geoguessr_ai_amd.training.train_eval_loop               <-> training/train_eval_loop.py

PyTorch-ROCm is used for device memory, streams and torch.distributed (RCCL) only.  There is no CPU fallback:
every op raises if ``libgg.so`` is missing or a tensor is not on the GPU.
"""
__version__ = "0.1.0"
