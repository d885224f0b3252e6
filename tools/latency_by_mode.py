"""dev: inference latency of TinyViT-21M-224 by batch size in the fp32 and fp32_split modes (small batches must not pay for the split GEMM's large tiles)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
for prec in ("fp32", "fp32_split"):
    m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision=prec).cuda().eval()
    for B in (4, 64, 256):
        x = torch.randn(B, 3, 224, 224, device="cuda")
        with torch.no_grad():
            for _ in range(3): m(x)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): m(x)
            torch.cuda.synchronize()
        print(prec, "batch", B, f"{(time.perf_counter() - t) * 100:.2f} ms per forward")
    del m
