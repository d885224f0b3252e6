"""dev: the reference default model (tiny_vit_21m_512: 32 x 32-token windows on the online-softmax kernels, 64 x 64 x ... token maps) forward + backward in the
fp32_split mode against the fp32 mode: embeddings and every weight gradient."""
import sys, torch
sys.path.insert(0, "/root/repo")
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
x = torch.randn(8, 3, 512, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
outs, grads = {}, {}
for prec in ("fp32", "fp32_split"):
    torch.manual_seed(5)
    m = TinyViTAdapter("tiny_vit_21m_512", pretrained=False, precision=prec).cuda().train()
    for p in m.parameters(): p.requires_grad_(True)
    y = m(x)
    y = y if torch.is_tensor(y) else getattr(y, "pooler_output", None) if getattr(y, "pooler_output", None) is not None else y.last_hidden_state
    (y.float() ** 2).mean().backward()
    outs[prec] = y.detach().float().cpu()
    grads[prec] = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None and p.dim() >= 2}
    del m; torch.cuda.empty_cache()
rel = lambda a, b: float((a - b).norm() / b.norm())
print("embedding rel", rel(outs["fp32_split"], outs["fp32"]))
w = sorted(((rel(grads["fp32_split"][n], g), n) for n, g in grads["fp32"].items() if float(g.norm()) > 0), reverse=True)[:3]
print("worst grads", w)
for prec in grads:
    bad = [n for n, g in grads[prec].items() if not torch.isfinite(g).all()]
    zero = [n for n, g in grads[prec].items() if float(g.norm()) == 0.0]
    print(prec, "tensors", len(grads[prec]), "non-finite", len(bad), bad[:3], "zero", len(zero), zero[:3])
