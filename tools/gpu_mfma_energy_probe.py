#!/usr/bin/env python3
"""What a FLOP costs the matrix pipe under the socket's power limit, by MFMA shape, by which operand consecutive MFMAs
share, and by which source the weights sit in.  Pure streams of independent MFMAs (one wave per SIMD, 256 workgroups,
register operands loaded once from random fp16 data: 'weights' ~ N(0, 0.05), 'activations' = mish(N(0, 1))), looped for
`seconds` per variant with the amdsmi sampler beside them: PFLOP/s, shader clock, socket power, pJ per FLOP above the
idle socket.      python tools/gpu_mfma_energy_probe.py [seconds]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "p3achygo_amd", "csrc", "asm"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_blockw_simcmp as T   # noqa: E402
from p3achygo_amd.power_sampler import PowerSampler   # noqa: E402


def order_pairs(na, nb, kind):
    if kind == "src0 held":
        return [(a, b) for a in range(na) for b in range(nb)]
    if kind == "src1 held":
        return [(a, b) for b in range(nb) for a in range(na)]
    # neither: a walk over all pairs in which consecutive pairs differ in both coordinates
    left = [(a, b) for a in range(na) for b in range(nb)]
    seq = [left.pop(0)]
    while left:
        for i, p in enumerate(left):
            if p[0] != seq[-1][0] and p[1] != seq[-1][1]:
                seq.append(left.pop(i))
                break
        else:
            seq.append(left.pop(0))
    return seq


def kernel(big, kind, iters, zero_c=False, lds_every=0, valu_per=0, dma_every=0):
    """lds_every: one ds_read_b128 (1 KiB per wave, random activations) every that many MFMAs; valu_per: that many plain VALU
    instructions (v_fma_f32 / v_pk_mul_f32 on live random values) after every MFMA; dma_every: one 1 KiB LDS-DMA piece per wave
    (global_load_lds_dwordx4 from the weights region, L2-resident) every that many MFMAs."""
    na, nb = (2, 3) if big else (4, 6)
    accw = 16 if big else 4
    L = ["k:", "\ts_load_dwordx4 s[4:7], s[0:1], 0x0", "\tv_lshlrev_b32 v1, 4, v0", "\ts_waitcnt lgkmcnt(0)"]
    for i in range(na):   # src0 fragments from the first region, src1 fragments from the second (64 KiB on)
        L.append(f"\tglobal_load_dwordx4 v[{20 + 4 * i}:{23 + 4 * i}], v1, s[4:5] offset:{i * 4096 % 4096}")
        L.append(f"\tv_add_u32 v1, 4096, v1")
    L.append("\tv_lshlrev_b32 v1, 4, v0")
    L.append("\tv_add_u32 v1, 65536, v1")
    for i in range(nb):
        L.append(f"\tglobal_load_dwordx4 v[{60 + 4 * i}:{63 + 4 * i}], v1, s[4:5]")
        L.append(f"\tv_add_u32 v1, 4096, v1")
    L.append("\ts_waitcnt vmcnt(0)")
    L.append("\tv_mov_b32 v2, 0")
    for r in range(na * nb * accw):
        L.append(f"\tv_accvgpr_write_b32 a{r}, v2")
    for r in range(224, 240):
        L.append(f"\tv_accvgpr_write_b32 a{r}, v2")
    # LDS image for the fragment reads: every lane's src1 fragments at lane * 16 + 4096 * i; VALU operands; DMA target
    L.append("\tv_lshlrev_b32 v5, 4, v0")
    for i in range(nb):
        L.append(f"\tds_write_b128 v5, v[{60 + 4 * i}:{63 + 4 * i}] offset:{4096 * i}")
    L.append("\ts_waitcnt lgkmcnt(0)")
    for r in range(100, 116):
        L.append(f"\tv_cvt_f32_f16 v{r}, v{60 + (r - 100) % 24}")
    L += ["\tv_lshrrev_b32 v6, 6, v0", "\ts_nop 3", "\tv_readfirstlane_b32 s20, v6", "\ts_nop 3", "\ts_lshl_b32 s21, s20, 10",
          "\ts_add_u32 s21, s21, 65536", "\tv_lshlrev_b32 v7, 4, v0", "\tv_and_b32 v7, 1023, v7", "\ts_mov_b32 m0, s21"]
    L.append(f"\ts_mov_b32 s30, {iters}")
    L.append(".Lloop:")
    pairs = order_pairs(na, nb, kind)
    count = 0
    reps = 96 // len(pairs) if not big else 48 // len(pairs)
    for _ in range(reps):
        for (a, b) in pairs:
            acc = (a * nb + b) * accw
            cin = f"a[{224}:{224 + accw - 1}]" if zero_c else f"a[{acc}:{acc + accw - 1}]"
            op = "v_mfma_f32_32x32x16_f16" if big else "v_mfma_f32_16x16x32_f16"
            L.append(f"\t{op} a[{acc}:{acc + accw - 1}], v[{20 + 4 * a}:{23 + 4 * a}], v[{60 + 4 * b}:{63 + 4 * b}], {cin}")
            count += 1
            if lds_every and count % lds_every == 0:
                k = (count // lds_every) % 8
                L.append(f"\tds_read_b128 v[{120 + 4 * k}:{123 + 4 * k}], v5 offset:{4096 * (k % nb)}")
                if k == 7:
                    L.append("\ts_waitcnt lgkmcnt(4)")
            for j in range(valu_per):
                k = (count * valu_per + j) % 16
                if j % 2 == 0:
                    L.append(f"\tv_fma_f32 v{160 + k}, v{100 + k}, v{100 + (k + 5) % 16}, v{100 + (k + 9) % 16}")
                else:
                    L.append(f"\tv_pk_mul_f32 v[{180 + 2 * (k % 8)}:{181 + 2 * (k % 8)}], v[{100 + 2 * (k % 8)}:{101 + 2 * (k % 8)}], v[{100 + 2 * ((k + 3) % 8)}:{101 + 2 * ((k + 3) % 8)}]")
            if dma_every and count % dma_every == 0:
                L.append("\tglobal_load_lds_dwordx4 v7, s[4:5]")
                if (count // dma_every) % 4 == 0:
                    L.append("\ts_waitcnt vmcnt(2)")
    L += ["\ts_sub_u32 s30, s30, 1", "\ts_cmp_lg_u32 s30, 0", "\ts_cbranch_scc1 .Lloop", "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_nop 15", "\ts_nop 15",
          "\tv_accvgpr_read_b32 v3, a0", "\tv_lshlrev_b32 v4, 2, v0", "\ts_lshl_b32 s13, s2, 10", "\tv_add_u32 v4, s13, v4",
          "\tglobal_store_dword v4, v3, s[6:7]", "\ts_waitcnt vmcnt(0)", "\ts_endpgm", ".Lfend:", "\t.size k, .Lfend-k"]
    per_iter = reps * len(pairs)
    return "\t.globl k\n\t.p2align 8\n\t.type k,@function\n" + "\n".join(L) + "\n", per_iter


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    gpu = T.Gpu()
    torch, hip = gpu.torch, gpu.hip
    rng = np.random.default_rng(3)
    n = 65536 // 2

    def mish(x):
        return x * np.tanh(np.log1p(np.exp(x)))

    data = {"weights": (rng.standard_normal(n) * 0.05).astype(np.float16), "activations": mish(rng.standard_normal(n)).astype(np.float16),
            "zeros": np.zeros(n, np.float16)}
    out = torch.zeros(256 * 256, dtype=torch.int32, device="cuda")
    smp = PowerSampler(0)
    idle = smp.read()
    print("idle (MHz, W):", idle, flush=True)
    nwg = 256
    for big in (False, True):
        for kind in ("src0 held", "src1 held", "neither held"):
            for roles in (("weights", "activations"), ("activations", "weights"), ("zeros", "zeros")):
                if roles[0] == "zeros" and kind != "src0 held":
                    continue
                iters = 6000 if not big else 6000
                text, per_iter = kernel(big, kind, iters)
                hs = T.assemble(text, "k")
                mod, fn = C.c_void_p(), C.c_void_p()
                buf = C.create_string_buffer(hs, len(hs))
                assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
                assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k") == 0
                td = torch.from_numpy(np.concatenate([data[roles[0]], data[roles[1]]]).view(np.int16).copy()).cuda()
                args = np.zeros(16, np.uint32)
                for i, t in ((0, td), (2, out)):
                    p = t.data_ptr()
                    args[i], args[i + 1] = p & 0xFFFFFFFF, p >> 32
                abuf = C.create_string_buffer(args.tobytes(), 64)
                size = C.c_size_t(64)
                extra = (C.c_void_p * 5)(C.c_void_p(1), C.cast(abuf, C.c_void_p), C.c_void_p(2), C.cast(C.pointer(size), C.c_void_p), C.c_void_p(3))
                for _ in range(3):
                    assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
                torch.cuda.synchronize()
                smp.start()
                t0 = time.perf_counter()
                launches = 0
                while time.perf_counter() - t0 < secs:
                    for _ in range(8):
                        assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
                    torch.cuda.synchronize()
                    launches += 8
                dt = time.perf_counter() - t0
                st = smp.stop() or {}
                flop = (32768 if big else 16384) * per_iter * iters * 4 * nwg * launches
                pf = flop / dt / 1e15
                p = st.get("socket_power_w_mean", 0.0)
                pj = (p - idle[1]) / (flop / dt) * 1e12 if pf > 0 else 0
                print(f"{'32x32x16' if big else '16x16x32'}  {kind:13s} src0 = {roles[0]:11s} src1 = {roles[1]:11s}  {pf:5.3f} PFLOP/s  "
                      f"({pf / 2.5:.2f} of peak)  clock {st.get('gfx_clock_mhz_mean', 0):5.0f} MHz  power {p:5.0f} W  "
                      f"{pj:5.2f} pJ/FLOP above idle  ppt {((st.get('limiter_residency') or {}).get('ppt', float('nan'))):.2f}", flush=True)
                hip.hipModuleUnload(mod)


def decomposition(secs):
    """the 16x16x32 stream with, one at a time and together, what the conv kernel does around its MFMAs, at about its ratios:
    a 1 KiB fragment read per 3 MFMAs, 2-3 VALU instructions per MFMA, a 1 KiB LDS-DMA piece per 24 MFMAs"""
    gpu = T.Gpu()
    torch, hip = gpu.torch, gpu.hip
    rng = np.random.default_rng(3)
    n = 65536 // 2
    x = rng.standard_normal(n)
    data = np.concatenate([(rng.standard_normal(n) * 0.05).astype(np.float16), (x * np.tanh(np.log1p(np.exp(x)))).astype(np.float16)])
    td = torch.from_numpy(data.view(np.int16).copy()).cuda()
    out = torch.zeros(256 * 256, dtype=torch.int32, device="cuda")
    smp = PowerSampler(0)
    idle = smp.read()
    print("idle (MHz, W):", idle, flush=True)
    nwg, iters = 256, 4000
    base = None
    for name, kw in (("MFMA alone", {}), ("+ fragment read / 3 MFMAs", {"lds_every": 3}), ("+ fragment read / 2 MFMAs", {"lds_every": 2}),
                     ("+ 2 VALU / MFMA", {"valu_per": 2}), ("+ 3 VALU / MFMA", {"valu_per": 3}), ("+ LDS-DMA piece / 24 MFMAs", {"dma_every": 24}),
                     ("+ read / 3, 2 VALU, DMA / 24", {"lds_every": 3, "valu_per": 2, "dma_every": 24}),
                     ("+ read / 3, 3 VALU, DMA / 24", {"lds_every": 3, "valu_per": 3, "dma_every": 24})):
        text, per_iter = kernel(False, "src0 held", iters, **kw)
        hs = T.assemble(text, "k")
        mod, fn = C.c_void_p(), C.c_void_p()
        buf = C.create_string_buffer(hs, len(hs))
        assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
        assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k") == 0
        args = np.zeros(16, np.uint32)
        for i, t in ((0, td), (2, out)):
            p = t.data_ptr()
            args[i], args[i + 1] = p & 0xFFFFFFFF, p >> 32
        abuf = C.create_string_buffer(args.tobytes(), 64)
        size = C.c_size_t(64)
        extra = (C.c_void_p * 5)(C.c_void_p(1), C.cast(abuf, C.c_void_p), C.c_void_p(2), C.cast(C.pointer(size), C.c_void_p), C.c_void_p(3))
        for _ in range(3):
            assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
        torch.cuda.synchronize()
        smp.start()
        t0 = time.perf_counter()
        launches = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(8):
                assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
            torch.cuda.synchronize()
            launches += 8
        dt = time.perf_counter() - t0
        st = smp.stop() or {}
        flop = 16384 * per_iter * iters * 4 * nwg * launches
        pf = flop / dt / 1e15
        p = st.get("socket_power_w_mean", 0.0)
        e = p / (flop / dt) * 1e12
        if base is None:
            base = e
        print(f"{name:34s} {pf:5.3f} PFLOP/s ({pf / 2.5:.2f} of peak)  clock {st.get('gfx_clock_mhz_mean', 0):5.0f} MHz  power {p:5.0f} W  "
              f"{e:5.3f} pJ/FLOP all in  ({e / base:4.2f} x the bare stream)", flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "decomposition":
        decomposition(float(sys.argv[2]) if len(sys.argv) > 2 else 1.5)
    else:
        main()
