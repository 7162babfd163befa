#!/usr/bin/env python3
"""Where a k_blockw block's cycles go: the diag kernel's section stamps for generator variants with pieces removed
(no ring DMA, no ring barriers, no epilogue fillers, other weaver budgets).  Timing only: the variants compute garbage.
    python tools/gpu_blockw_phaseprobe.py [L]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p3achygo_amd", "csrc", "asm"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockw_gen as G          # noqa: E402
import blockw_ref as R          # noqa: E402
import gpu_blockw_simcmp as T   # noqa: E402


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    npos, nblk, nwg = 1024, 3, 256
    gpu = T.Gpu()
    torch, hip = gpu.torch, gpu.hip
    rng = np.random.default_rng(1)
    blocks = [R.random_block(rng, L) for _ in range(nblk)]
    ws, prm = zip(*[R.pack_block(W, bn, L) for (W, bn) in blocks])
    tw = torch.from_numpy(np.concatenate(ws).view(np.int16).copy()).cuda()
    tp = torch.from_numpy(np.concatenate(prm).copy()).cuda()
    tx = (torch.randn(npos * 256 * 361, device="cuda") * 0.5).half()
    ts = torch.zeros(8 * 16 * 4 * 24, dtype=torch.int64, device="cuda")
    variants = [("full", {}), ("nofiller", {"nofiller": 1}), ("nodma", {"nodma": 1}), ("nodma nofiller", {"nodma": 1, "nofiller": 1}),
                ("nobarrier", {"nobarrier": 1}), ("nodma nobarrier nofiller", {"nodma": 1, "nobarrier": 1, "nofiller": 1}),
                ("budget 4", {"budget": 4}), ("budget 12", {"budget": 12}), ("budget 16", {"budget": 16}),
                ("nodma nobarrier", {"nodma": 1, "nobarrier": 1}), ("nowrite", {"nowrite": 1}), ("noread", {"noread": 1, "nodma": 1, "nobarrier": 1}),
                ("noread nofiller", {"noread": 1, "nodma": 1, "nobarrier": 1, "nofiller": 1}),
                ("noaccread", {"noaccread": 1}), ("notrans", {"notrans": 1}), ("noperm", {"noperm": 1}),
                ("nowrite noaccread noperm", {"nowrite": 1, "noaccread": 1, "noperm": 1}),
                ("synth 400", {"synthfill": 400}), ("synth 800", {"synthfill": 800}), ("synth 1200", {"synthfill": 1200}),
                ("synth 800 nodma nobarrier", {"synthfill": 800, "nodma": 1, "nobarrier": 1}),
                ("synth 800 noread", {"synthfill": 800, "nodma": 1, "nobarrier": 1, "noread": 1}),
                ("synth 800 noread sync", {"synthfill": 800, "noread": 1}),
                ("synth 800 budget 16", {"synthfill": 800, "budget": 16}),
                ("synth 800 budget 12", {"synthfill": 800, "budget": 12}),
                ("synth 800 budget 32", {"synthfill": 800, "budget": 32})]
    if len(sys.argv) > 2:
        variants = [v for v in variants if v[0] in sys.argv[2:]]
    for name, opts in variants:
        g = G.BlockGen(L, True)
        g.opts = opts
        g.kernel("k")
        hs = T.assemble(g.e.text(), "k")
        mod, fn = C.c_void_p(), C.c_void_p()
        buf = C.create_string_buffer(hs, len(hs))
        assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
        assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k") == 0
        args = np.zeros(16, np.uint32)
        for i, t in ((0, tx), (2, tw), (4, tp), (10, ts)):
            p = t.data_ptr()
            args[i], args[i + 1] = p & 0xFFFFFFFF, p >> 32
        args[6], args[7], args[8] = npos, nblk, nwg
        abuf = C.create_string_buffer(args.tobytes(), 64)
        size = C.c_size_t(64)
        extra = (C.c_void_p * 5)(C.c_void_p(1), C.cast(abuf, C.c_void_p), C.c_void_p(2), C.cast(C.pointer(size), C.c_void_p), C.c_void_p(3))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for rep in range(4):
            if rep == 1:
                ev0.record()
            assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 3
        st = ts.cpu().numpy().reshape(8, 16, 4, 24)[:, 1]     # second block of the first position of workgroups 0..7
        n = g.nstamp
        d = np.diff(st[:, :, :n], axis=2).reshape(-1, n - 1)
        med = np.median(d, axis=0).astype(int)
        rt = (st[:, :, 22] - st[:, :, 21]).astype(np.float64) * 10e-9          # seconds (100 MHz)
        cyc = (st[:, :, 23] - st[:, :, 0]).astype(np.float64)
        ghz = float(np.median(cyc / rt)) / 1e9
        print(f"{name:26s} launch {ms:7.3f} ms  block {int(np.median(st[:, :, n - 1] - st[:, :, 0])):7d} cycles  clock {ghz:.2f} GHz  sections {med.tolist()}", flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    main()
