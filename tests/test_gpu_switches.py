"""Every development switch of libgg (GG_*; read only when GG_DEV_SWITCHES is set) selects a kernel or schedule that is ALSO a product path --
the fallback the default schedule takes for shapes its fast kernels do not cover (channel counts that are not multiples of 4, single
windows, K above the prologue table, ...).  This file runs the same training-step parity check the default path gets (tools/switch_parity.py:
TinyViT-5M step vs the CPU oracle, fp32 and bf16 modes, per-tensor gradients) with the switches set, in groups, one subprocess per group
(the library reads a switch once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = {
    # every fusion off, every multi-column / multi-window / resident / ring form replaced by its plain fallback
    "plain_fallbacks": dict(GG_NO_FUSE_DW_S1="1", GG_NO_FUSE_DW_S2="1", GG_NO_FUSE_BNBWD="1", GG_NO_BNGEMM="1", GG_NO_PRO="1", GG_F32_NO_FUSE="1", GG_NO_LN_COLSUM="1",
                            GG_DW_NO_MULTI="1", GG_DW_F32_NO_MULTI="1", GG_ATTN_SMALL="0", GG_ATTN_FLASH_NO_RES="1", GG_ATTN_DQ_QS1="1",
                            GG_GEMM_F32_SB="0", GG_GEMM_F32_ROWS_EPI="0", GG_ATTN_NO_DS_SCRATCH="1", GG_ATTN_NO_FUSED_BWD="1", GG_ATTN_FWD_NO_TAIL="1", GG_GEMM_F32_PRO_RING="0", GG_GEMM_F32_NO_W96="1",
                            GG_ATTN_NO_SPLIT="1", GG_GEMM_DMA="0", GG_GEMM_TILE="n", GG_GEMM_F32_NO_SMALL="1", GG_SWITCH_PARITY_FROZEN="1"),
    # the older kernel generation: LDS-tiled depthwise, one-column fused forward, register-staged fp32 GEMM, 64-byte-run stores
    "older_kernels": dict(GG_DW_TILED="1", GG_FUSE_DW="1", GG_NO_FUSE_BNBWD_EPI="1", GG_GEMM_F32_RING="0", GG_GEMM_F32_NO_PERSIST="1",
                          GG_GEMM_F32_DEBUG="128", GG_DW_NO_MULTI_PLAIN="1", GG_DW_NO_MULTI_BWD="1", GG_DW_S2_TILED="1", GG_GEMM_TILE="w",
                          GG_ATTN_NO_FUSED_BWD="1", GG_ATTN_NO_SPLIT="1", GG_GEMM_F32_NO_SPLITK="1", GG_GEMM_F32_SMALL_MAX="64", GG_SWITCH_PARITY_FROZEN="1"),       # (two-pass attention backward with the dS hand-off: the form for windows beyond 256 tokens)
    # the single-pass attention backward with one wave per key strip (no cooperative tail strip): the form every window whose strip count is not 4 n + 1 takes
    "attention_no_tail": dict(GG_ATTN_FUSED_NO_TAIL="1", GG_ATTN_NO_SPLIT="1", GG_SWITCH_PARITY_FROZEN="1"),
    # the f32-MFMA window attention (the form head dim 64 and windows other than 7 x 7 / 12 x 12 / 14 x 14 take) with everything else at its default; the
    # 16-bit LDS-DMA GEMM in its 256 x 256 geometry wherever the shape allows it
    "attention_f32_mfma_gemm16_256": dict(GG_ATTN_NO_SPLIT="1", GG_GEMM_DMA_TILE="256"),
    # the default schedule with every parameter trainable (unfused BatchNorm / weight-gradient paths of all stages) and under the freeze policy
    "default_unfrozen": dict(),
    "default_frozen": dict(GG_SWITCH_PARITY_FROZEN="1"),
}


@pytest.mark.parametrize("group", sorted(GROUPS))
def test_parity_under_switch_group(group):
    env = dict(os.environ, GG_DEV_SWITCHES="1", **GROUPS[group])
    env.pop("GG_PRECISION", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "switch_parity.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stdout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count("-> ok") == 2


def test_switches_are_inert_without_the_dev_gate():
    """Without GG_DEV_SWITCHES a GG_* switch must not change anything: a timing-experiment value that would corrupt results (GG_GEMM_F32_DEBUG=2
    drops the GEMM's result stores) still gives the correct step."""
    env = dict(os.environ, GG_GEMM_F32_DEBUG="2", GG_GEMM_DEBUG="2", GG_SWITCH_PARITY_FROZEN="1")
    env.pop("GG_DEV_SWITCHES", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "switch_parity.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("-> ok") == 2, (r.stdout[-2000:], r.stderr[-2000:])
