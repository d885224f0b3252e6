"""N>1 path on CPU: two gloo ranks shard a batch, sum-all-reduce the flat gradient buffers exactly like the GPU
ranks do over RCCL, and end up with identical parameters equal to the single-process result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeBackbone(torch.nn.Module):
    """Stands in for TinyVitBackbone's flat-storage protocol (no GPU needed): params are views of one flat buffer."""

    def __init__(self):
        super().__init__()
        self._flat = torch.arange(32, dtype=torch.float32) / 10
        self._flat_grad = torch.zeros(32)
        self.w = torch.nn.Parameter(self._flat[0:8].view(2, 4))
        self.frozen = torch.nn.Parameter(self._flat[8:16], requires_grad=False)
        self.v = torch.nn.Parameter(self._flat[16:32])
        self._params = {"w": self.w, "frozen": self.frozen, "v": self.v}

    flat_params = property(lambda self: self._flat)

    def flat_grads(self):
        return self._flat_grad

    def trainable_ranges(self):
        return [(0, 8), (16, 32)]

    def mark_params_dirty(self):
        pass


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geoguessr_ai_amd.optim import AdamW
    from geoguessr_ai_amd.training.train_eval_loop import _batches
    bb = _FakeBackbone()
    opt = AdamW(bb, lr=1e-2)
    data = dict(x=torch.arange(24, dtype=torch.float32).view(12, 2))
    seen = []
    for batch in _batches(data, 3, True, 5, rank, world):
        seen += batch["x"][:, 0].tolist()
        bb.flat_grads()[0:8] += batch["x"].sum()            # "backward": gradient depends on this rank's shard
        bb.flat_grads()[16:32] += batch["x"].mean()
    opt.allreduce_grads()
    g = bb.flat_grads().clone()
    out.put((rank, seen, g))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, seen0, g0), (_, seen1, g1) = res
    assert sorted(seen0 + seen1) == [float(v) for v in range(0, 24, 2)]       # ranks partition the epoch, no overlap
    assert torch.equal(g0, g1)                                                 # summed over ranks
    assert float(g0[8:16].abs().sum()) == 0.0                                  # the frozen range is never exchanged
    data = torch.arange(24, dtype=torch.float32).view(12, 2)
    assert abs(float(g0[0]) - float(data.sum())) < 1e-3                        # == single-process gradient sum
