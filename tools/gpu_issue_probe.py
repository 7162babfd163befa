#!/usr/bin/env python3
"""Issue cost of single instructions beside MFMA 16x16x32 on one wave per SIMD (the weaver's budget table).

A generated loop of [MFMA + fillers] x 96 per iteration, 200 iterations, four waves per workgroup; s_memtime around the
loop.  Prints cycles per MFMA gap for each filler mix.   python tools/gpu_issue_probe.py [workgroups]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p3achygo_amd", "csrc", "asm"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import blockw_gen as G          # noqa: E402
import gpu_blockw_simcmp as T   # noqa: E402

FILL = {
    "none": [],
    "fma x1": ["v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "fma x2": ["v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "fma x2 3-same-bank": ["v_fma_f32 v{a4}, v{a4}, v136, v140", "v_fma_f32 v{d4}, v{d4}, v136, v140"],
    "fma x2 2-same-bank": ["v_fma_f32 v{a4}, v{a4}, v136, v141", "v_fma_f32 v{d4}, v{d4}, v136, v141"],
    "fma x2 dst-only-same": ["v_fma_f32 v{a4}, v137, v138, v139", "v_fma_f32 v{d4}, v137, v138, v139"],
    "fma x1 3-same-bank": ["v_fma_f32 v{a4}, v{a4}, v136, v140"],
    "ldsv + fma x2": ["ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "ldsa + fma x2": ["ds_read_b128 a[{q0}:{q3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "ldsv/3 + fma x2": ["?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "ldsa/3 + fma x2": ["?3 ds_read_b128 a[{q0}:{q3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "ldsv/3": ["?3 ds_read_b128 v[{r0}:{r3}], v{z}"],
    "ldsv/3 + exp": ["?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_exp_f32 v{a}, v{b}"],
    "ldsv/3 + fma x1": ["?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "ldsv/3 used + fma x2": ["?3 ds_read_b128 v[{u0}:{u3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "W15 ldsv/3 + fma x1": ["@s_waitcnt lgkmcnt(15)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "W15 ldsv/3 + fma x2": ["@s_waitcnt lgkmcnt(15)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "W2 ldsv/3 + fma x1": ["@s_waitcnt lgkmcnt(2)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "W1 ldsv/3 + fma x1": ["@s_waitcnt lgkmcnt(1)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "W0 ldsv/3 + fma x1": ["@s_waitcnt lgkmcnt(0)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "W1 ldsv/3": ["@s_waitcnt lgkmcnt(1)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}"],
    "W0 ldsv/3": ["@s_waitcnt lgkmcnt(0)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}"],
    "W1 lds(a+v)/3 + fma x1": ["@s_waitcnt lgkmcnt(1)", "?3 ds_read_b128 v[{r0}:{r3}], v{z}", "?3 ds_read_b128 a[{q0}:{q3}], v{z}", "v_fma_f32 v{a}, v{b}, v{c}, v{a}"],
    "fma x3": ["v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}", "v_fma_f32 v{e}, v{b}, v{c}, v{e}"],
    "fma x4": ["v_fma_f32 v{a}, v{b}, v{c}, v{a}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}", "v_fma_f32 v{e}, v{b}, v{c}, v{e}", "v_fma_f32 v{f}, v{b}, v{c}, v{f}"],
    "add x2": ["v_add_f32 v{a}, 2.0, v{a}", "v_add_f32 v{d}, 2.0, v{d}"],
    "exp x1": ["v_exp_f32 v{a}, v{b}"],
    "exp x2": ["v_exp_f32 v{a}, v{b}", "v_exp_f32 v{d}, v{c}"],
    "rcp x1": ["v_rcp_f32 v{a}, v{b}"],
    "exp + fma": ["v_exp_f32 v{a}, v{b}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"],
    "accread x1": ["v_accvgpr_read_b32 v{a}, a{g}"],
    "accread x2": ["v_accvgpr_read_b32 v{a}, a{g}", "v_accvgpr_read_b32 v{d}, a{h}"],
    "mixlo x1": ["v_fma_mixlo_f16 v{a}, v{b}, v{c}, 0"],
    "mixlo x2": ["v_fma_mixlo_f16 v{a}, v{b}, v{c}, 0", "v_fma_mixlo_f16 v{d}, v{b}, v{c}, 0"],
    "mix_f32 x2": ["v_fma_mix_f32 v{a}, v{b}, v{c}, v{c} op_sel:[0,0,0] op_sel_hi:[1,0,0]", "v_fma_mix_f32 v{d}, v{b}, v{c}, v{c} op_sel:[1,0,0] op_sel_hi:[1,0,0]"],
    "cvt_pk x1": ["v_cvt_pk_f16_f32 v{a}, v{b}, v{c}"],
    "cvt_pk x2": ["v_cvt_pk_f16_f32 v{a}, v{b}, v{c}", "v_cvt_pk_f16_f32 v{d}, v{b}, v{c}"],
    "permswap x1": ["v_permlane16_swap_b32 v{a}, v{d}"],
    "pk_fma x1": ["v_pk_fma_f32 v[{a2}:{a3}], v[{b2}:{b3}], v[{b2}:{b3}], v[{a2}:{a3}]"],
    "ds_read x1": ["ds_read_b128 v[{r0}:{r3}], v{z}"],
    "ds_write x1": ["ds_write_b128 v{z}, v[{r0}:{r3}]"],
    "mov exec x2": ["s_mov_b64 exec, -1", "s_mov_b64 exec, -1"],
    "s_nop 0": ["s_nop 0"],
    "s_nop 1": ["s_nop 1"],
    "s_add x2": ["s_add_u32 s40, s40, 1", "s_addc_u32 s41, s41, 0"],
    "waitcnt": ["s_waitcnt lgkmcnt(0)"],
}
NM = 96


def kernel(fill, dep_chain=False, NM=NM, iters=200, aloc="v", bloc="v", same_b=False, big=False, nfill=1):
    L = ["k:", "\ts_load_dwordx2 s[4:5], s[0:1], 0x0", "\tv_mov_b32 v250, 0", "\tv_lshlrev_b32 v251, 4, v0"]
    for r in range(100, 140):
        L.append(f"\tv_mov_b32 v{r}, 0x3f000000")
    for r in range(20, 60):
        L.append(f"\tv_mov_b32 v{r}, 0x38003400")
    for r in range(224, 256):
        L.append(f"\tv_accvgpr_write_b32 a{r}, v20")
    L.append("\tds_write_b128 v251, v[20:23]")
    L.append("\ts_waitcnt lgkmcnt(0)")
    L.append(f"\ts_mov_b32 s30, {iters}")
    L.append("\ts_memtime s[8:9]")
    L.append("\ts_waitcnt lgkmcnt(0)")
    L.append(".Lloop:")
    for m in range(NM):
        acc = (m % 24) * 4
        for t in fill:
            if t.startswith("@"):
                L.append("\t" + t[1:])
        ao = (224 if aloc == "a" else 20) + 4 * (m % 4)
        bo = (240 if bloc == "a" else 40) + (0 if same_b else 4 * (m % 4))
        if big:
            acc = (m % 12) * 16
            L.append(f"\tv_mfma_f32_32x32x16_f16 a[{acc}:{acc + 15}], {aloc}[{ao}:{ao + 3}], {bloc}[{bo}:{bo + 3}], a[{acc}:{acc + 15}]")
        else:
            L.append(f"\tv_mfma_f32_16x16x32_f16 a[{acc}:{acc + 3}], {aloc}[{ao}:{ao + 3}], {bloc}[{bo}:{bo + 3}], a[{acc}:{acc + 3}]")
        k = m % 8
        regs = dict(a4=100 + 4 * (k % 4), d4=116 + 4 * (k % 4), a=100 + k, d=108 + k, e=116 + k, f=124 + k, b=132, c=133, g=200 + (m % 16), h=216 + (m % 16), z=251,
                    r0=60 + 4 * (m % 4), r3=63 + 4 * (m % 4), a2=100 + 2 * k, a3=101 + 2 * k, b2=134, b3=135)
        regs.update(q0=224 + 4 * ((m + 2) % 4), q3=227 + 4 * ((m + 2) % 4), u0=40 + 4 * ((m + 2) % 4), u3=43 + 4 * ((m + 2) % 4))
        for t in fill * nfill:
            if t.startswith("@"):
                continue
            if t.startswith("?3 "):
                if m % 3:
                    continue
                t = t[3:]
                if "u0" in t:
                    L.append("\ts_waitcnt lgkmcnt(0)")
            L.append("\t" + t.format(**regs))
    L.append("\ts_sub_u32 s30, s30, 1")
    L.append("\ts_cmp_lg_u32 s30, 0")
    if NM * 8 * (1 + len(fill)) < 100000:
        L.append("\ts_cbranch_scc1 .Lloop")
    else:
        L += ["\ts_cbranch_scc0 .Lout", "\ts_getpc_b64 s[36:37]", ".Lpc:", "\ts_add_u32 s36, s36, .Lloop-.Lpc", "\ts_addc_u32 s37, s37, -1",
              "\ts_setpc_b64 s[36:37]", ".Lout:"]
    L.append("\ts_waitcnt lgkmcnt(0)")
    L.append("\ts_nop 15")
    L.append("\ts_memtime s[10:11]")
    L.append("\ts_waitcnt lgkmcnt(0)")
    L += ["\ts_sub_u32 s12, s10, s8", "\tv_mov_b32 v252, s12", "\tv_lshlrev_b32 v253, 2, v0",
          "\ts_lshl_b32 s13, s2, 10", "\tv_add_u32 v253, s13, v253",
          "\tglobal_store_dword v253, v252, s[4:5]", "\ts_waitcnt vmcnt(0)", "\ts_endpgm", ".Lfend:", "\t.size k, .Lfend-k"]
    return "\t.globl k\n\t.p2align 8\n\t.type k,@function\n" + "\n".join(L) + "\n"


def dma_kernel(mode, nfma=5, iters=100):
    """loop of 36 MFMAs 32x32x16 (+ nfma plain VALU each) with three 1 KiB LDS-DMA pieces per wave per iteration:
    mode "none" no DMA; "burst" all three right after an s_barrier; "burst_nobar"; "spread" one per 12 MFMAs;
    "stagger" one per 12 MFMAs at a wave-dependent MFMA index (wave w: after MFMA 3 w + 12 k)"""
    L = ["k:", "\ts_load_dwordx4 s[4:7], s[0:1], 0x0", "\tv_lshlrev_b32 v251, 4, v0", "\tv_lshrrev_b32 v1, 6, v0",
         "\ts_nop 3", "\tv_readfirstlane_b32 s20, v1", "\ts_nop 4", "\ts_lshl_b32 s21, s20, 10", "\ts_add_u32 s21, s21, 65536"]
    for r in range(100, 140):
        L.append(f"\tv_mov_b32 v{r}, 0x3f000000")
    for r in range(20, 60):
        L.append(f"\tv_mov_b32 v{r}, 0x38003400")
    L += ["\ts_waitcnt lgkmcnt(0)", "\ts_mov_b64 s[22:23], s[6:7]", f"\ts_mov_b32 s30, {iters}", "\ts_memtime s[8:9]", "\ts_waitcnt lgkmcnt(0)", ".Lloop:"]

    def dma(k):
        return [f"\ts_add_u32 m0, s21, {4096 * k}", "\ts_nop 0", "\tglobal_load_lds_dwordx4 v251, s[22:23]",
                "\ts_add_u32 s22, s22, 4096", "\ts_addc_u32 s23, s23, 0"]
    if mode in ("burst", "burst_nobar"):
        L.append("\ts_waitcnt vmcnt(3)")
        if mode == "burst":
            L.append("\ts_barrier")
        for k in range(3):
            L += dma(k)
    for m in range(36):
        acc = (m % 12) * 16
        L.append(f"\tv_mfma_f32_32x32x16_f16 a[{acc}:{acc + 15}], v[{20 + 4 * (m % 4)}:{23 + 4 * (m % 4)}], v[{40 + 4 * (m % 4)}:{43 + 4 * (m % 4)}], a[{acc}:{acc + 15}]")
        for f in range(nfma):
            L.append(f"\tv_fma_f32 v{100 + (m * nfma + f) % 32}, v132, v133, v{100 + (m * nfma + f) % 32}")
        if mode == "spread" and m % 12 == 5:
            L.append("\ts_waitcnt vmcnt(2)")
            L += dma(m // 12)
        if mode == "stagger" and m % 12 in (1, 4, 7, 10):
            w = (1, 4, 7, 10).index(m % 12)
            L += [f"\ts_cmp_lg_u32 s20, {w}", f"\ts_cbranch_scc1 .Lskip{m}", "\ts_waitcnt vmcnt(2)"] + dma(m // 12) + [f".Lskip{m}:"]
    L += ["\ts_cmp_lt_u64 s[22:23], s[16:17]" if False else "\ts_sub_u32 s30, s30, 1", "\ts_cmp_lg_u32 s30, 0", "\ts_cbranch_scc1 .Lloop",
          "\ts_waitcnt vmcnt(0) lgkmcnt(0)", "\ts_nop 15", "\ts_memtime s[10:11]", "\ts_waitcnt lgkmcnt(0)",
          "\ts_sub_u32 s12, s10, s8", "\tv_mov_b32 v252, s12", "\tv_lshlrev_b32 v253, 2, v0", "\ts_lshl_b32 s13, s2, 10",
          "\tv_add_u32 v253, s13, v253", "\tglobal_store_dword v253, v252, s[4:5]", "\ts_waitcnt vmcnt(0)", "\ts_endpgm", ".Lfend:",
          "\t.size k, .Lfend-k"]
    return "\t.globl k\n\t.p2align 8\n\t.type k,@function\n" + "\n".join(L) + "\n"


def main():
    nwg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    if len(sys.argv) > 2 and sys.argv[2] == "dma":
        gpu = T.Gpu()
        torch, hip = gpu.torch, gpu.hip
        out = torch.zeros(nwg * 256, dtype=torch.int32, device="cuda")
        src = torch.zeros(4 << 20, dtype=torch.uint8, device="cuda")      # weights source: 100 iterations x 12 KiB per workgroup, shared
        for nf in (0, 3, 5):
            for mode in ("none", "burst", "burst_nobar", "spread", "stagger"):
                hs = T.assemble(dma_kernel(mode, nf), "k")
                mod, fn = C.c_void_p(), C.c_void_p()
                buf = C.create_string_buffer(hs, len(hs))
                assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
                assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k") == 0
                args = np.zeros(16, np.uint32)
                p, q = out.data_ptr(), src.data_ptr()
                args[0], args[1], args[2], args[3] = p & 0xFFFFFFFF, p >> 32, q & 0xFFFFFFFF, q >> 32
                abuf = C.create_string_buffer(args.tobytes(), 64)
                size = C.c_size_t(64)
                extra = (C.c_void_p * 5)(C.c_void_p(1), C.cast(abuf, C.c_void_p), C.c_void_p(2), C.cast(C.pointer(size), C.c_void_p), C.c_void_p(3))
                torch.cuda.synchronize()
                for rep in range(3):
                    assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
                torch.cuda.synchronize()
                cyc = out.cpu().numpy().reshape(nwg, 256)[:, ::64].astype(np.float64)
                per = np.median(cyc) / 100
                print(f"fma x{nf} {mode:12s} cycles per 36-MFMA group {per:8.1f}  (MFMA alone 1152)   per DMA {(per - 1152 - 0) / 3:6.1f}", flush=True)
                hip.hipModuleUnload(mod)
        return
    gpu = T.Gpu()
    torch = gpu.torch
    hip = gpu.hip
    out = torch.zeros(nwg * 256, dtype=torch.int32, device="cuda")
    cases = list(FILL.items())
    if len(sys.argv) > 2 and sys.argv[2] == "operands":
        f2 = FILL["fma x2"]
        cases = []
        for al in "va":
            for bl in "va":
                for sb in (False, True):
                    for nm, fl in (("none", []), ("fma x2", f2), ("exp x1", FILL["exp x1"]), ("accread x2", FILL["accread x2"])):
                        cases.append((f"A={al} B={bl} sameB={int(sb)} {nm}", ("ops", al, bl, sb, fl)))
    if len(sys.argv) > 2 and sys.argv[2] == "banks":
        cases = [(n, f) for n, f in FILL.items() if "bank" in n or "same" in n or n in ("none", "fma x2")]
    if len(sys.argv) > 2 and sys.argv[2] == "lds":
        cases = [(n, f) for n, f in FILL.items() if n.startswith("W") or n in ("none", "ldsv/3 + fma x1")]
    if len(sys.argv) > 2 and sys.argv[2] == "big":
        one = ["v_fma_f32 v{a}, v{b}, v{c}, v{a}"]
        cases = [(f"32x32x16 + fma x{k}", ("big", one, k)) for k in (0, 2, 4, 5, 6, 7, 8)]
        cases += [("32x32x16 + exp x2 + fma x2", ("big", ["v_exp_f32 v{a}, v{b}", "v_fma_f32 v{d}, v{b}, v{c}, v{d}"], 2)),
                  ("32x32x16 + exp x3", ("big", ["v_exp_f32 v{a}, v{b}"], 3)),
                  ("32x32x16 + lds + wait + fma x4", ("big", ["@s_waitcnt lgkmcnt(15)", "ds_read_b128 v[{r0}:{r3}], v{z}"] + one * 4, 1)),
                  ("32x32x16 + mixlo x3", ("big", ["v_fma_mixlo_f16 v{a}, v{b}, v{c}, 0"], 3)),
                  ("32x32x16 + cvt_pk x3 + mul x3", ("big", ["v_cvt_pk_f16_f32 v{a}, v{b}, v{c}", "v_mul_f32 v{d}, v{b}, v{c}"], 3))]
    if len(sys.argv) > 2 and sys.argv[2] == "icache":
        # straight-line code of growing size, same number of MFMAs in all: does instruction fetch keep up with one wave per SIMD?
        cases = [(f"code {nm * 8 // 1024:4d} KiB", (nm, 19200 // nm * 10)) for nm in (96, 1200, 2400, 4800, 9600, 19200)]
        cases += [(f"code {nm * 16 // 1024:4d} KiB +fma", (nm, 19200 // nm * 10, ["v_fma_f32 v{a}, v{b}, v{c}, v{a}"])) for nm in (96, 2400, 9600)]
        cases += [(f"code {nm * 24 // 1024:4d} KiB +fma x2", (nm, 19200 // nm * 10, FILL["fma x2"])) for nm in (96, 2400, 9600)]
        cases += [(f"code {nm * 12 // 1024:4d} KiB +add x1 (4-byte)", (nm, 19200 // nm * 10, ["v_add_f32 v{a}, 2.0, v{a}"])) for nm in (96, 2400, 9600)]
    for name, fill in cases:
        global NM
        iters = 200
        if isinstance(fill, tuple) and fill[0] == "big":
            hs = T.assemble(kernel(fill[1], big=True, nfill=fill[2]), "k")
            NM_eff = NM * 2      # counted in 16-cycle units: a 32x32x16 is two gaps' worth of FLOPs
        elif isinstance(fill, tuple) and fill[0] == "ops":
            hs = T.assemble(kernel(fill[4], aloc=fill[1], bloc=fill[2], same_b=fill[3]), "k")
            NM_eff = NM
        elif isinstance(fill, tuple):
            NMl, iters = fill[0], fill[1]
            fl = fill[2] if len(fill) > 2 else []
            hs = T.assemble(kernel(fl, NM=NMl, iters=iters), "k")
            NM_eff = NMl
        else:
            hs = T.assemble(kernel(fill), "k")
            NM_eff = NM
        mod, fn = C.c_void_p(), C.c_void_p()
        buf = C.create_string_buffer(hs, len(hs))
        assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
        assert hip.hipModuleGetFunction(C.byref(fn), mod, b"k") == 0
        args = np.zeros(16, np.uint32)
        p = out.data_ptr()
        args[0], args[1] = p & 0xFFFFFFFF, p >> 32
        abuf = C.create_string_buffer(args.tobytes(), 64)
        size = C.c_size_t(64)
        extra = (C.c_void_p * 5)(C.c_void_p(1), C.cast(abuf, C.c_void_p), C.c_void_p(2), C.cast(C.pointer(size), C.c_void_p), C.c_void_p(3))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for rep in range(3):
            if rep == 1:
                ev0.record()
            assert hip.hipModuleLaunchKernel(fn, nwg, 1, 1, 256, 1, 1, 0, None, None, extra) == 0
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 2
        cyc = out.cpu().numpy().reshape(nwg, 256)[:, ::64].astype(np.float64)
        per = np.median(cyc) / (iters * NM_eff)
        tf = nwg * 4 * iters * NM_eff * 16384 / (ms * 1e-3) / 1e12
        print(f"{name:14s} cycles per MFMA gap {per:6.2f}   (+{per - 16:5.2f})   wall {ms:7.3f} ms  {tf:7.0f} TFLOP/s  clock {np.median(cyc) / (ms * 1e-3) / 1e9:.2f} GHz", flush=True)
        hip.hipModuleUnload(mod)


if __name__ == "__main__":
    main()
