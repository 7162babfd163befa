"""k_blockw (p3achygo_amd/csrc/asm/blockw_gen.py: the C = 256 btl block as generated gfx950 assembly, opt-in with
P3HIP_BLOCKW=1) without a GPU: the generated instruction stream runs in the functional emulator (asm/sim.py) on one
workgroup and must reproduce a numpy restatement of the bottleneck block (model.py:372-425: x + conv1x1(conv3x3...(conv1x1(x))),
every conv = conv(mish(bn(.))), model.py:276-292) within fp16 rounding; the emitter's own bookkeeping — counted waits, the
loop-head state, register map — is exercised by generating every variant."""
import os
import sys

import numpy as np
import pytest

ASM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "p3achygo_amd", "csrc", "asm")
sys.path.insert(0, ASM)


def _run(L, nblk, npos=1, seed=1):
    import blockw_gen as G
    import blockw_ref as R
    import sim
    g = G.BlockGen(L, False)
    g.kernel("k")
    rng = np.random.default_rng(seed)
    blocks = [R.random_block(rng, L) for _ in range(nblk)]
    xs = [(rng.standard_normal((256, 361)) * 0.5).astype(np.float16) for _ in range(npos)]
    ws, prm = zip(*[R.pack_block(W, bn, L) for (W, bn) in blocks])
    ws, prm = np.concatenate(ws), np.concatenate(prm)
    assert len(ws) * 2 == nblk * g.ngran * 4096 and len(prm) == nblk * g.prm_floats
    mem = sim.Mem()
    ax = mem.add(np.concatenate([R.x_to_device(x).reshape(-1) for x in xs]))
    aw, ap = mem.add(ws), mem.add(prm)
    karg = np.zeros(16, np.uint32)
    for i, a in ((0, ax), (2, aw), (4, ap)):
        karg[i], karg[i + 1] = a & 0xFFFFFFFF, a >> 32
    karg[6], karg[7], karg[8] = npos, nblk, 1          # one workgroup walks every position
    ak = mem.add(karg)
    s = sim.Sim(g.e.text(), "k", mem, ak, 0)
    s.run()
    out = mem.array(ax, np.float16, npos * 256 * 361).reshape(npos, -1)
    worst = 0.0
    for p, x in enumerate(xs):
        ref = x
        for (W, bn) in blocks:
            ref = R.block_ref(ref, W, bn, L)
        got = R.x_from_device(out[p]).astype(np.float32)
        assert not np.isnan(got).any()
        worst = max(worst, float(np.abs(got - ref.astype(np.float32)).max()))
    return worst


def test_one_block_one_inner_layer():
    # fp16 outputs of magnitude ~2: one or two units in the last place (the BN scale is folded into the fp16 weights)
    assert _run(1, 1) <= 4e-3


def test_two_blocks_two_positions_walk_the_flat_loop():
    """position and block form one runtime loop: weights and parameters of block 1, the wrap back to block 0 for the
    second position, the ring's twelve slots re-entered at every block"""
    assert _run(1, 2, npos=2, seed=3) <= 6e-3


@pytest.mark.parametrize("L", [2, 3])
def test_more_inner_layers(L):
    assert _run(L, 1, seed=10 + L) <= 6e-3


def test_every_variant_generates_and_keeps_its_invariants():
    """the generator's assertions (register ranges, offsets, loop-head vector-memory state, granule count) hold for every
    kernel the library ships, and the accumulators / fragments / temporaries do not overlap"""
    import blockw_gen as G
    for L in (1, 2, 3):
        for diag in (False, True):
            g = G.BlockGen(L, diag)
            g.kernel("k")
            assert g.e.stats["mfma"] == (32 + 72 * L) * 12
    assert G.ACC["B"] + 96 <= G.FA and G.FA + 32 <= G.INIT and G.INIT + 32 <= 256
    assert G.R0 + 48 <= G.P0 and G.P0 + 32 <= G.XBUF[0] and G.XBUF[0] + 48 <= G.XBUF[1] and G.XBUF[1] + 48 <= G.T0
    assert G.LDS_BYTES <= 163840
