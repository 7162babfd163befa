"""Register allocation of the fused block kernels, read from the compiler's own assembly.

k_block<256, 128, ...> runs at 255-256 of the 256 VGPRs two waves per SIMD leave a wave, and what the allocator
parks in scratch depends on code far away from the block loop: with the broadcast dense fused into the launch's tail
(kernels.hip, tail_dense) two harmless-looking edits of that tail (zeroing only the act buffer's halo slots;
laundering a lane index) made it keep A1 — the 48-register activated half that waits across the reduce conv's first
K slice — in scratch INSIDE the block loop: 19 scratch loads and as many stores per block, +7 % on the launch
(gpurun_out/dfuse_ab3.log), with every parity test still green.  This test is the guard: it compiles kernels.hip to
assembly for gfx950 (no GPU needed) and checks that
  * no k_block instantiation without fused broadcast convs touches scratch at all, and
  * the C = 256 instantiations with them (BC) touch scratch only outside the block loop (head, tail and prologue
    run once per position; the block loop runs 4-6 times and is where the time goes)."""
import collections
import hashlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "p3achygo_amd", "csrc")
SOURCES = ["kernels.hip", "kernels.h", "conv_core.h", "conv16.h"]
HEAD_MFMAS = 384          # the fused conv_last: four segments of 4 k32 steps x 24 MFMAs


def _assembly():
    h = hashlib.sha256()
    for f in SOURCES:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    out = os.path.join(ROOT, "build", "kernels_gfx950_%s.s" % h.hexdigest()[:16])
    if not os.path.exists(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        for old in os.listdir(os.path.dirname(out)):
            if old.startswith("kernels_gfx950_") and old.endswith(".s"):
                os.remove(os.path.join(os.path.dirname(out), old))
        r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                            os.path.join(CSRC, "kernels.hip"), "-o", out + ".tmp"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        os.replace(out + ".tmp", out)
    return open(out).read()


def _kernels(asm):
    """{(C, CB, kind, L, NW, BC): [(mfma index, loop depth, instruction)] of scratch ops, total MFMAs}"""
    res = {}
    for m in re.finditer(r"^_ZN2p37k_blockILi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d)ELb(\d)EEEvNS_9BlockArgsE:", asm, re.M):
        key = tuple(int(g) for g in m.groups())
        body = asm[m.start():asm.index(".Lfunc_end", m.start())].split("\n")
        mf, depth, ops = 0, 0, []
        for line in body:
            t = line.strip()
            if t.startswith(".LBB"):
                d = re.search(r"Depth=(\d+)", t)
                depth = int(d.group(1)) if d else 0
            if "v_mfma" in t:
                mf += 1
            if "scratch_" in t:
                ops.append((mf, depth, t.split(";")[0].strip()))
        res[key] = (ops, mf)
    return res


@pytest.fixture(scope="module")
def kernels():
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    return _kernels(_assembly())


def test_block_kernels_without_fused_broadcast_convs_use_no_scratch(kernels):
    plain = {k: v for k, v in kernels.items() if k[5] == 0}
    assert len(plain) >= 12
    for key, (ops, _) in plain.items():
        assert not ops, (key, ops[:5])


def test_block_loop_of_the_c256_kernels_with_fused_broadcast_convs_touches_no_scratch(kernels):
    seen = 0
    for key, (ops, total) in kernels.items():
        C, CB, kind, L, NW, BC = key
        if not (C == 256 and BC == 1):
            continue
        seen += 1
        block = kernels[(C, CB, kind, L, NW, 0)][1]          # MFMAs of one block in the code = the plain kernel's
        assert total >= HEAD_MFMAS + block
        # inside the block loop: after the head's last MFMA, before the tail's first, in a block of loop depth >= 2
        # (depth 1 = the position loop: head, tail and the code between them run once per position)
        inside = [o for o in ops if HEAD_MFMAS < o[0] < HEAD_MFMAS + block and o[1] >= 2]
        # the loop's top and bottom share their MFMA index with the head's end / the tail's start: there the count
        # tells a handful of once-per-position reloads from a parked 48-register value
        edge = collections.Counter(o[0] for o in ops if o[0] in (HEAD_MFMAS, HEAD_MFMAS + block))
        assert not inside, (key, inside[:8])
        assert all(n <= 12 for n in edge.values()), (key, dict(edge))
    assert seen >= 4
