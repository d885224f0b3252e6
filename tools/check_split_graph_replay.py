"""dev: whole-encoder inference in the fp32_split mode at a graph-captured size (64 images): eager first call, captured second, replays -- bit-identical outputs."""
import os, sys, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd import _lib as L
m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision=os.environ.get("P", "fp32_split")).cuda().eval()
import os
x = torch.randn(int(os.environ.get("B", "64")), 3, 224, 224, device="cuda")
outs = []
with torch.no_grad():
    for i in range(4):
        y = m(x); y = y if torch.is_tensor(y) else (y.pooler_output if getattr(y, "pooler_output", None) is not None else y.last_hidden_state)
        outs.append(y.cpu()); del y          # (the caching allocator hands the next call the same output address: the graph key repeats)
torch.cuda.synchronize()
cap, rep, eag = C.c_int64(), C.c_int64(), C.c_int64()
L.lib().gg_graph_stats(C.byref(cap), C.byref(rep), C.byref(eag))
print("graphs captured", cap.value, "replays", rep.value, "eager", eag.value, "identical", all(torch.equal(outs[0], o) for o in outs[1:]), float(outs[0].abs().mean()))
