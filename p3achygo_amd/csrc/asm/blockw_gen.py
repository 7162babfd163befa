#!/usr/bin/env python3
"""Generator of k_blockw: the bottleneck residual blocks of a C = 256 / C_b = 128 trunk, one wave per SIMD.

    python blockw_gen.py OUT.s            (kernels k_blockw_L1, _L2, _L3 and their _diag twins)

Arithmetic spec: BottleneckResidualConvBlock, /root/reference/python/model.py:372-425 (x + conv1x1(... conv3x3(
... conv1x1(x)))), every conv = conv(mish(bn(.))), model.py:276-292), net shapes model_config.py:98-105.  The HIP
kernel k_block<256,128,btl> (kernels.hip) computes the same thing with eight waves and 256 registers each; DESIGN.md
section 4 (round 4) explains why this form exists: with two waves per SIMD the BN + mish epilogues (VALU) never ran
under MFMA work, 44 % of a block's cycles.

Shape.  One 256-thread workgroup (four waves, one per SIMD, 512 registers each: 256 VGPR + 256 AGPR) owns a position.
Wave w owns board-row tile pairs t = w, w + 4, w + 8 (32 padded rows each) and ALL 128 output channels of a layer, as
two accumulator sets in AGPRs: set A = channels 0..63, set B = 64..127 (96 registers each).  The act buffer in LDS
(one slot of 128 channels per padded board point, conv_core.h Geo<1,128,3>) is updated IN PLACE, half by half:
    3x3 layer l:   phase 1 (A, cin lo)   phase 2 (B, cin lo)   phase 3 (A, cin hi)   phase 4 (B, cin hi)
    under phase 1: epilogue of set B of layer l-1 -> channels hi      (hi is next read by phase 3)
    under phase 4: epilogue of set A of layer l   -> channels lo      (lo was last read by phase 2)
so BN + mish of 96 values per lane ride in the issue slots the MFMAs leave free (an MFMA 16x16x32 occupies the
vector issue for 8 of its 16 cycles), stage by stage, eight values in flight.  Weights stream HBM/L2 -> LDS ring by
LDS-DMA in 4 KiB granules (64 output channels x 32 k), twelve slots, one barrier per group of granules.
A block's input is always x in HBM (fetched in the 16-byte piece layout, activated, written to LDS), its output x'
goes to HBM: the block loop carries no register state, and position / block form one flat runtime loop.
"""
import math
import sys

from emitter import (A, S, V, Bundle, Emitter, Ins, accread, ds_read128, ds_write128, gload128, gstore128, mfma, misc,
                     permswap16, rtxt, salu, trans, valu)

# ---- geometry (conv_core.h Geo<1, 128, 3>) ----------------------------------------------------
SLOTB = 272                     # bytes per act slot: 128 channels fp16 + 16 B pad
SROW = 20                       # padded row stride
PADTOP = 21
PSLOTS = PADTOP + 379 + PADTOP
ACT_BYTES = (PSLOTS * SLOTB + 15) // 16 * 16            # 114,512
GRAN = 4096                     # weight granule: 64 couts x 32 k fp16
NSLOT = 12
RING0 = ACT_BYTES
LDS_BYTES = RING0 + NSLOT * GRAN                        # 163,664
assert LDS_BYTES <= 163840
NLOC = 361
CBLK_BYTES = NLOC * 16          # bytes of one 8-channel block of one position in x
POS_BYTES = 32 * CBLK_BYTES     # C = 256

# ---- register map -----------------------------------------------------------------------------
# AGPR
ACC = {"A": 0, "B": 96}         # acc quad (ct, j) of set s: a[ACC[s] + (ct * 6 + j) * 4 ...]
FA = 192                        # weight fragments: fa[buf][ct] = a[FA + (buf * 4 + ct) * 4 ...]


def acc(s, ct, j):
    return A(ACC[s] + (ct * 6 + j) * 4, 4)


def fa(buf, ct):
    return A(FA + (buf * 4 + ct) * 4, 4)


# VGPR
V_TID = 0
V_DMAOFF = 1        # wave * 1024 + lane * 16: this lane's 16 bytes of a granule (global source and LDS destination)
V_ARING = 2         # RING0 + lane * 16: this lane's piece of weight fragment ct = 0 of ring slot 0
V_BROW = 3          # [3] LDS address of fragment R[b][0] at ky = 0, q32 = 0
V_WROW = 6          # [3] LDS address of this lane's 16-byte piece of its row of tile pair b (chunk q >> 1)
V_XOFF = 9          # [3] byte offset of this lane's piece of pair b inside channel block (q >> 1) of the position
V_QOFF = 12         # q * 16: byte offset of this lane's four channels inside a 16-channel cout tile (fp32 parameters)
V_POFF = 13         # (q >> 1) * 32: byte offset of this lane's eight channels inside a 16-channel tile (fp32 parameters)
V_C1 = 14           # -2 ln 2
V_C2 = 15           # ln 2
V_ZERO = 16         # [4] zeros
R0 = 20             # [12] activation fragments, 4 registers each


def rfrag(i):
    return V(R0 + 4 * i, 4)


P0 = 68             # [64] BN parameters, two banks of 32
XBUF = (132, 208)   # [2][48] x pieces of a quarter: (ct, b) -> XBUF[k] + (ct * 3 + b) * 4
T0 = 180            # [8] temporaries
W0 = 188            # [8]
O0 = 196            # [4] packed results of a (pair, cout tile)
TMP = 200           # [8] scratch

# SGPR
S_KARG = 0          # [2]
S_WG = 2
S_X = 4             # [2] x
S_WS = 6            # [2] weight stream of the launch's blocks
S_PRM = 8           # [2] parameter table
S_NPOS = 10
S_NBLK = 11
S_NWG = 12
S_STAMP = 14        # [2] diag stamps
S_POS = 16
S_BLK = 17
S_XP = 18           # [2] x of this position
S_XQ = 20           # [2] scratch: a channel block of this position (stores)
S_XL = 42           # [2] scratch: a channel block of this position (loads)
S_DMA = 22          # [2] next granule to fetch (wave's piece excluded: in V_DMAOFF)
S_PB = 24           # [2] parameters of this block
S_PQ = 26           # [2] scratch: a parameter row
S_RINGW = 28        # RING0 + wave * 1024: LDS address of the wave's piece of slot 0 (M0 = S_RINGW + slot * 4096)
S_OK = 30           # [3][2] lanes whose row of tile pair b is on the board
S_TMP = 36          # [4]
S_WAVE = 40
S_T0 = 44           # [2 * 24] diag stamps of one block

LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def f32hex(x):
    import struct
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


class BlockGen:
    def __init__(self, L, diag=False, dump_at=None):
        self.L = L
        self.diag = diag
        self.dump_at = dump_at            # debugging: at stamp point k dump LDS and registers to the stamps buffer, then end
        self.stop = None                  # debugging: "prologue" / "store" / "xloads" / "prm" / "dma" / "entry": end there
        self.e = Emitter()
        self.ngran = 32 + 72 * L          # granules of one block: reduce 16, layers 72 each, expand 16
        self.pbank = 0
        self.nstamp = 0
        # parameter table of one block (floats), every row pre-multiplied by log2(e) on the host (engine.cpp):
        #   bn0 scale[256] shift[256] | layer j = 1..L+1: scale[128] shift[128]
        self.prm_floats = 512 + 256 * (L + 1)

    # ---- ring ---------------------------------------------------------------------------------
    def dma_issue(self, g):
        """LDS-DMA of granule g of this block (or of the next block at the block's end) into slot g % NSLOT"""
        slot = g % NSLOT
        return Bundle([
            salu("s_add_u32", "m0", S(S_RINGW), slot * GRAN),
            misc("s_nop 0"),
            Ins(f"global_load_lds_dwordx4 {rtxt(V(V_DMAOFF))}, {rtxt(S(S_DMA, 2))}", "dma", [], [V(V_DMAOFF), S(S_DMA, 2)], 16, ("g", g)),
            salu("s_add_u32", S(S_DMA), S(S_DMA), GRAN),
            salu("s_addc_u32", S(S_DMA + 1), S(S_DMA + 1), 0),
        ])

    def ring_sync(self, need):
        """main-stream items: the granules in `need` have landed for every wave, older slots are free: refill them."""
        items = []
        tags = {("g", g) for g in need}
        g0 = min(need)

        def wait_and_barrier(e, tags=tags):
            e.wait_vm_tags(tags)
            e.raw("\ts_barrier")
        items.append(("call", wait_and_barrier))
        while self.next_dma < min(g0 + NSLOT, self.ngran):
            items.append(self.dma_issue(self.next_dma))
            self.next_dma += 1
        return items

    # ---- fragment addressing ---------------------------------------------------------------------
    @staticmethod
    def b_off(ky, s, chunk4):
        """immediate offset of activation fragment R[.][s] of kernel row ky, 16-byte chunk group chunk4 (k32 index)"""
        return (ky * SROW + s) * SLOTB + chunk4 * 64

    @staticmethod
    def a_off(g, ct):
        return (g % NSLOT) * GRAN + ct * 1024

    # ---- 3x3 phase -------------------------------------------------------------------------------
    def phase3x3(self, g_base, sset, half, first):
        """main-stream items of one phase: set `sset` ("A"/"B") over input channels half (0 = lo, 1 = hi): 18 k32 steps in
        the order (ky, q32, kx) - three taps of a kernel row share their activation fragments (conv16.h
        conv_segment16_3x3) - weights: granules g_base .. g_base + 17.  `first`: the accumulators start from zero."""
        items = []
        steps = [(ky, q, kx) for ky in range(3) for q in range(2) for kx in range(3)]

        def afetch(u, cts):
            g = g_base + u
            return [ds_read128(fa(u % 2, ct), V(V_ARING), self.a_off(g, ct)) for ct in cts]

        def bfetch(grp, b, s):
            ky, q = divmod(grp, 2)
            return ds_read128(rfrag(4 * b + s), V(V_BROW + b), self.b_off(ky, s, 2 * half + q))
        # prologue: first group's weights have landed (sync), fragments of group 0
        items += self.ring_sync(range(g_base, g_base + 3))
        items += afetch(0, range(4))
        for b in range(3):
            for s in range(4):
                items.append(bfetch(0, b, s))
        for u, (ky, q, kx) in enumerate(steps):
            grp = u // 3
            last_grp = grp == 5
            nxt = u + 1 < 18
            for j in range(6):
                b, par = divmod(j, 2)
                for ct in range(4):
                    cin = 0 if (first and u == 0) else acc(sset, ct, j)
                    items.append(mfma(acc(sset, ct, j), fa(u % 2, ct), rfrag(4 * b + kx + par), cin))
                if par == 0:
                    if not last_grp:
                        items.append(bfetch(grp + 1, b, kx))           # the even tile was the last user of R[b][kx]
                    if j == 0 and nxt:
                        if (u + 1) % 3 == 0:
                            items += self.ring_sync(range(g_base + u + 1, g_base + u + 4))
                        items += afetch(u + 1, (0, 1))
                else:
                    if kx == 2 and not last_grp:
                        items.append(bfetch(grp + 1, b, 3))
                    if j == 1 and nxt:
                        items += afetch(u + 1, (2, 3))
        return items

    # ---- 1x1 phase: NH half-steps of 24 MFMAs ---------------------------------------------------------
    def phase1x1(self, g_base, plan):
        """plan: list of k32 steps; a step = (chunk4, [(set, first), ...]): the six centre-tap fragments of 16-byte chunk
        group chunk4 are multiplied with one granule of weights per listed set, in that order."""
        items = []
        halfsteps = []
        for si, (chunk4, sets) in enumerate(plan):
            for (sset, first) in sets:
                halfsteps.append((si, chunk4, sset, first))
        ng = len(halfsteps)
        sync_every = 4

        def afetch(h, cts):
            return [ds_read128(fa(h % 2, ct), V(V_ARING), self.a_off(g_base + h, ct)) for ct in cts]

        def bfetch(si, j):
            chunk4 = plan[si][0]
            b, par = divmod(j, 2)
            # centre tap: ky = 1, kx = 1 -> s = 1 + parity
            return ds_read128(rfrag(6 * (si % 2) + j), V(V_BROW + b), self.b_off(1, 1 + par, chunk4))
        items += self.ring_sync(range(g_base, g_base + min(sync_every, ng)))
        items += afetch(0, range(4))
        for j in range(6):
            items.append(bfetch(0, j))
        for h, (si, chunk4, sset, first) in enumerate(halfsteps):
            new_step_next = h + 1 < ng and halfsteps[h + 1][0] != si
            for j in range(6):
                for ct in range(4):
                    cin = 0 if first else acc(sset, ct, j)
                    items.append(mfma(acc(sset, ct, j), fa(h % 2, ct), rfrag(6 * (si % 2) + j), cin))
                if h + 1 < ng:
                    if j == 0:
                        if (h + 1) % sync_every == 0:
                            items += self.ring_sync(range(g_base + h + 1, g_base + min(h + 1 + sync_every, ng)))
                        items += afetch(h + 1, (0, 1))
                    if j == 1:
                        items += afetch(h + 1, (2, 3))
                    if new_step_next and j >= 2:
                        # the next step's fragments go to the other buffer, whose last user was the step before this one
                        items.append(bfetch(si + 1, j - 2))
                        if j == 5:
                            items.append(bfetch(si + 1, 4))
                            items.append(bfetch(si + 1, 5))
        return items

    # ---- fillers ---------------------------------------------------------------------------------
    def prm_row(self, float_off):
        """items that point S_PQ at a row of this block's parameter table"""
        return [salu("s_add_u32", S(S_PQ), S(S_PB), float_off * 4), salu("s_addc_u32", S(S_PQ + 1), S(S_PB + 1), 0)]

    def mish8(self, t, w):
        """stage by stage over eight values: t[i] = log2(e) * bn(v) in, u in w[i] out (mish = t * u, applied by the caller)"""
        it = []
        for i in range(8):
            it.append(trans("v_exp_f32", V(w + i), V(t + i)))
        for i in range(8):
            it.append(valu("v_add_f32", V(TMP + i), 2.0, V(w + i)))
        for i in range(8):
            it.append(valu("v_fma_f32", V(w + i), V(w + i), V(TMP + i), 2.0))
        for i in range(8):
            it.append(trans("v_rcp_f32", V(w + i), V(w + i)))
        for i in range(8):
            it.append(valu("v_fma_f32", V(w + i), V(w + i), V(V_C1), V(V_C2)))
        return it

    def pack8(self, t, w, o):
        """o[0..3] = fp16 of t[i] * w[i]: (o0, o1) = values 0..3, (o2, o3) = values 4..7.  The four low halves first, then
        the four high halves: a partial register write is never followed directly by the other half's (which reads it)"""
        it = []
        for i in (0, 2, 4, 6, 1, 3, 5, 7):
            op = "v_fma_mixlo_f16" if i % 2 == 0 else "v_fma_mixhi_f16"
            ins = valu(op, V(o + i // 2), V(t + i), V(w + i), 0, note="dstsel")
            if i % 2:
                ins.src |= {("v", o + i // 2)}    # mixhi keeps the low half
            it.append(ins)
        return it

    def epi_loads(self, sset, prm_off):
        """items: this lane's BN parameters of accumulator set `sset` (scale and shift of its 4 x 4 channels) -> the
        set's parameter bank.  prm_off: float offset of the layer's scale row in the block's table (shift row 128 on)."""
        bank = P0 + (0 if sset == "A" else 32)
        it = self.prm_row(prm_off + (0 if sset == "A" else 64))
        for ct in range(4):
            it.append(gload128(V(bank + 8 * ct, 4), V(V_QOFF), S(S_PQ, 2), ct * 64, "prm"))
            it.append(gload128(V(bank + 8 * ct + 4, 4), V(V_QOFF), S(S_PQ, 2), 512 + ct * 64, "prm"))
        return it

    def epi_layer(self, sset, dst_half):
        """filler items: BN + mish of accumulator set `sset` -> fp16 -> act buffer channels half dst_half
        (parameters: epi_loads, issued a phase earlier)"""
        it = []
        bank = P0 + (0 if sset == "A" else 32)
        for b in range(3):
            for ct in range(4):
                sc, sh = bank + 8 * ct, bank + 8 * ct + 4
                for i in range(8):
                    j, k = 2 * b + i // 4, i % 4
                    it.append(accread(V(T0 + i), A(ACC[sset] + (ct * 6 + j) * 4 + k)))
                for i in range(8):
                    it.append(valu("v_fma_f32", V(T0 + i), V(T0 + i), V(sc + i % 4), V(sh + i % 4)))
                it += self.mish8(T0, W0)
                it += self.pack8(T0, W0, O0)
                it.append(permswap16(V(O0), V(O0 + 2)))
                it.append(permswap16(V(O0 + 1), V(O0 + 3)))
                it.append(self.masked(b, ds_write128(V(V_WROW + b), V(O0, 4), dst_half * 128 + ct * 32)))
        return it

    def masked(self, b, ins):
        return Bundle([misc(f"s_mov_b64 exec, {rtxt(S(S_OK + 2 * b, 2))}"), ins, misc("s_mov_b64 exec, -1")])

    def xq_point(self, quarter, ct, sreg=S_XQ):
        """items: sreg = channel block (quarter * 8 + 2 ct) of this position"""
        off = (quarter * 8 + 2 * ct) * CBLK_BYTES
        return [salu("s_add_u32", S(sreg), S(S_XP), off), salu("s_addc_u32", S(sreg + 1), S(S_XP + 1), 0)]

    def x_loads(self, quarter, buf, tag):
        """items: this lane's twelve 16-byte pieces of x quarter `quarter` -> X buffer `buf`"""
        it = []
        for ct in range(4):
            it += self.xq_point(quarter, ct, S_XL)
            for b in range(3):
                it.append(gload128(V(XBUF[buf] + (ct * 3 + b) * 4, 4), V(V_XOFF + b), S(S_XL, 2), 0, tag))
        return it

    def act_write(self, buf, dst_half):
        """items: the activated pieces in X buffer `buf` -> act buffer half (on-board rows only)"""
        return [self.masked(b, ds_write128(V(V_WROW + b), V(XBUF[buf] + (ct * 3 + b) * 4, 4), dst_half * 128 + ct * 32))
                for ct in range(4) for b in range(3)]

    def act_loads(self, quarter, ct):
        """items: bn0 parameters of this lane's eight channels of (quarter, cout tile ct) -> act bank (quarter * 4 + ct) % 2"""
        bank = P0 + 32 + 16 * ((quarter * 4 + ct) % 2)
        it = self.prm_row(quarter * 64 + ct * 16)
        for k in range(2):      # scale (k = 0: first four channels), shift
            it.append(gload128(V(bank + 4 * k, 4), V(V_POFF), S(S_PQ, 2), 16 * k, "prm"))
            it.append(gload128(V(bank + 8 + 4 * k, 4), V(V_POFF), S(S_PQ, 2), 1024 + 16 * k, "prm"))
        return it

    def act_quarter(self, quarter, buf, dst_half, write=True, then=None):
        """filler items: the fetched pieces of x quarter `quarter` (X buffer `buf`) -> mish(bn0(.)), in place;
        write: -> act buffer half as each piece is done (otherwise act_write follows later).  The parameters of
        (quarter, 0) were requested by the caller (act_loads); each cout tile requests the next one's, the last
        one those of (then, 0)."""
        X0 = XBUF[buf]
        it = []
        for ct in range(4):
            bank = P0 + 32 + 16 * ((quarter * 4 + ct) % 2)
            if ct + 1 < 4:
                it += self.act_loads(quarter, ct + 1)
            elif then is not None:
                it += self.act_loads(then, 0)
            for b in range(3):
                x = X0 + (ct * 3 + b) * 4
                for i in range(8):
                    hi = i % 2
                    it.append(valu("v_fma_mix_f32", V(T0 + i), V(x + i // 2), V(bank + i), V(bank + 8 + i),
                                   mods=f" op_sel:[{hi},0,0] op_sel_hi:[1,0,0]"))
                it += self.mish8(T0, W0)
                it += self.pack8(T0, W0, x)
                if write:
                    it.append(self.masked(b, ds_write128(V(V_WROW + b), V(x, 4), dst_half * 128 + ct * 32)))
        return it

    def epi_expand(self, quarter, sset, buf):
        """filler items: x' = acc + x (fp16) for output channels quarter*64.. -> HBM (the X registers hold the residual
        pieces, fetched by x_loads(quarter)); expand epilogue of conv16.h epilogue_store16"""
        X0 = XBUF[buf]
        it = []
        for ct in range(4):
            it += self.xq_point(quarter, ct)
            for b in range(3):
                x = X0 + (ct * 3 + b) * 4
                for i in range(8):
                    j, k = 2 * b + i // 4, i % 4
                    it.append(accread(V(T0 + i), A(ACC[sset] + (ct * 6 + j) * 4 + k)))
                # piece halves -> (my four channels of tile 2b, of tile 2b + 1)
                it.append(permswap16(V(x), V(x + 2)))
                it.append(permswap16(V(x + 1), V(x + 3)))
                for i in (0, 2, 4, 6, 1, 3, 5, 7):
                    op = "v_fma_mixlo_f16" if i % 2 == 0 else "v_fma_mixhi_f16"
                    ins = valu(op, V(O0 + i // 2), V(T0 + i), 1.0, V(x + i // 2), mods=f" op_sel:[0,0,{i % 2}] op_sel_hi:[0,0,1]",
                               note="dstsel")
                    if i % 2:
                        ins.src |= {("v", O0 + i // 2)}
                    it.append(ins)
                it.append(permswap16(V(O0), V(O0 + 2)))
                it.append(permswap16(V(O0 + 1), V(O0 + 3)))
                it.append(self.masked(b, gstore128(V(V_XOFF + b), V(O0, 4), S(S_XQ, 2), 0, "xst")))
        return it

    # ---- diagnostics ------------------------------------------------------------------------------
    def stamp(self):
        """main-stream item: s_memtime into the block's stamp registers (diag build only)"""
        if self.dump_at is not None:
            k = self.nstamp
            self.nstamp += 1
            return [("call", self.dump_and_end)] if k == self.dump_at else []
        if not self.diag:
            return []
        k = self.nstamp
        self.nstamp += 1
        assert k < 24

        def f(e, k=k):
            e.wait_lgkm_all()
            e.raw(f"\ts_memtime {rtxt(S(S_T0 + 2 * k, 2))}")
            e.raw("\ts_waitcnt lgkmcnt(0)")
        return [("call", f)]

    def dump_and_end(self, e, parts=("vgpr", "sgpr", "agpr", "lds")):
        """debugging (tools/gpu_blockw_simcmp.py): the workgroup's LDS and every wave's registers -> the stamps buffer,
        [LDS 163,840 B][wave w: v0..v255, a0..a255 as 512 x 64 dwords], then the program ends.  v248..v255 are scratch."""
        e.wait_vm_all()
        e.wait_lgkm_all()
        e.raw("\ts_barrier")
        # registers first (v248.. are clobbered afterwards): wave w at 163840 + w * 131072, register r at r * 256 + lane * 4
        # (the dump goes behind position 0 of the x buffer, which the harness allocates large enough: the stamps pointer
        # is not trusted here — it is one of the things being debugged; the SGPRs go out too, as "register" v248..)
        e.raw(f"\ts_add_u32 s{S_STAMP}, s{S_X}, {POS_BYTES}")
        e.raw(f"\ts_addc_u32 s{S_STAMP + 1}, s{S_X + 1}, 0")
        e.raw(f"\ts_mul_i32 s{S_TMP}, s{S_WAVE}, 131072")
        e.raw(f"\ts_add_u32 s{S_TMP}, s{S_TMP}, 163840")
        e.raw(f"\ts_add_u32 s{S_TMP + 2}, s{S_STAMP}, s{S_TMP}")
        e.raw(f"\ts_addc_u32 s{S_TMP + 3}, s{S_STAMP + 1}, 0")
        e.raw("\tv_mov_b32 v250, v0")
        e.raw("\tv_and_b32 v251, 63, v0")
        e.raw("\tv_lshlrev_b32 v251, 2, v251")            # lane * 4
        for r in range(248):
            if "vgpr" in parts:
                e.raw(f"\tglobal_store_dword v251, v{r}, s[{S_TMP + 2}:{S_TMP + 3}] offset:{(r * 256) % 4096}")
            if (r * 256) % 4096 == 3840:
                e.raw(f"\ts_add_u32 s{S_TMP + 2}, s{S_TMP + 2}, 4096")
                e.raw(f"\ts_addc_u32 s{S_TMP + 3}, s{S_TMP + 3}, 0")
        # SGPRs 0..63 in the slot of "register" 248, 64..95 in that of 249 (lane i = SGPR i)
        for base, reg in (((0, 248), (64, 249)) if "sgpr" in parts else ()):
            e.raw("\tv_mov_b32 v252, 0")
            e.raw("\ts_nop 3")
            for i in range(64 if base == 0 else 32):
                if base + i in (S_STAMP, S_STAMP + 1):
                    continue
                e.raw(f"\tv_writelane_b32 v252, s{base + i}, {i}")
            e.raw("\ts_nop 1")
            e.raw(f"\tglobal_store_dword v251, v252, s[{S_TMP + 2}:{S_TMP + 3}] offset:{(reg * 256) % 4096}")
            e.raw("\ts_nop 1")
        # (248 = 15.5 pages: the pointer stands at register 240's page; registers 250..255 are skipped)
        e.raw(f"\ts_add_u32 s{S_TMP + 2}, s{S_TMP + 2}, 4096")
        e.raw(f"\ts_addc_u32 s{S_TMP + 3}, s{S_TMP + 3}, 0")
        for r in range(256):
            if "agpr" in parts:
                e.raw(f"\tv_accvgpr_read_b32 v252, a{r}")
                e.raw("\ts_nop 1")
                e.raw(f"\tglobal_store_dword v251, v252, s[{S_TMP + 2}:{S_TMP + 3}] offset:{(r * 256) % 4096}")
                e.raw("\ts_nop 1")
            if (r * 256) % 4096 == 3840:
                e.raw(f"\ts_add_u32 s{S_TMP + 2}, s{S_TMP + 2}, 4096")
                e.raw(f"\ts_addc_u32 s{S_TMP + 3}, s{S_TMP + 3}, 0")
        # LDS: thread t copies bytes [t * 16 + 4096 * i, +16), i = 0 .. 39
        e.raw("\tv_lshlrev_b32 v253, 4, v250")
        e.raw(f"\ts_mov_b64 s[{S_TMP + 2}:{S_TMP + 3}], s[{S_STAMP}:{S_STAMP + 1}]")
        for i in range(40 if "lds" in parts else 0):
            e.raw("\tds_read_b128 v[244:247], v253")
            e.raw("\ts_waitcnt lgkmcnt(0)")
            e.raw(f"\tglobal_store_dwordx4 v253, v[244:247], s[{S_TMP + 2}:{S_TMP + 3}]")
            e.raw("\ts_waitcnt vmcnt(0)")
            e.raw("\tv_add_u32 v253, 4096, v253")
        e.raw("\ts_waitcnt vmcnt(0)")
        e.raw("\ts_endpgm")

    # ---- the block ------------------------------------------------------------------------------
    def block_body(self):
        L = self.L
        e = self.e
        main = []
        self.next_dma = NSLOT          # granules 0..11 were issued at the end of the previous block
        self.nstamp = 0
        g = 0
        main += self.stamp()
        # -- the block's input: x quarters 0, 1 (requested at the block's entry, with the first parameters) -> act buffer
        # lo, hi; quarters 2, 3 are requested as the buffers come free and activated under / behind the reduce's first half
        main += self.act_quarter(0, 0, 0, then=1)
        main += self.x_loads(2, 0, "x2")
        main += self.act_quarter(1, 1, 1, then=2)
        main += self.x_loads(3, 1, "x3")
        main.append(Ins("", "wait_lds_writes"))
        main += self.stamp()
        main.append(("start", "act2", self.act_quarter(2, 0, 0, write=False, then=3)))
        plan = [(c, [("A", c == 0), ("B", c == 0)]) for c in range(4)]
        main += self.phase1x1(g, plan)
        g += 8
        main.append(("flush", "act2"))
        main += self.stamp()
        # lo and hi are free once every wave has left the K loop above
        main.append(("call", lambda e: e.raw("\ts_barrier")))
        main += self.act_write(0, 0)
        main += self.epi_loads("A", 512)             # the reduce epilogue's parameters, a K loop ahead
        main += self.act_quarter(3, 1, 1)
        main += self.epi_loads("B", 512)
        main.append(Ins("", "wait_lds_writes"))
        main += self.stamp()
        plan = [(c, [("A", False), ("B", False)]) for c in range(4)]
        main += self.phase1x1(g, plan)
        g += 8
        main += self.stamp()
        # -- reduce epilogue: set A exposed (lo is read by the first phase), set B under phase 1 -----------------
        main.append(("call", lambda e: e.raw("\ts_barrier")))       # every wave is done reading the act buffer
        main += self.epi_layer("A", 0)
        main.append(Ins("", "wait_lds_writes"))
        main += self.stamp()
        pendingB = self.epi_layer("B", 1)
        for l in range(1, L + 1):
            prm_next = 512 + 256 * l
            # phase 1 (A, lo) with the previous layer's set B epilogue
            main.append(("start", "epiB", pendingB))
            main += self.phase3x3(g, "A", 0, True)
            g += 18
            # phase 2 starts set B afresh: the previous layer's set B must have been read out
            main.append(("flush", "epiB"))
            main += self.phase3x3(g, "B", 0, True)
            g += 18
            # hi must be complete (every wave's set-B epilogue written) before phase 3 reads it
            main.append(Ins("", "wait_lds_writes"))
            # this layer's epilogue parameters: both banks are free now
            main += self.epi_loads("A", prm_next)
            main += self.epi_loads("B", prm_next)
            main += self.phase3x3(g, "A", 1, False)
            g += 18
            main.append(("start", "epiA", self.epi_layer("A", 0)))
            main += self.phase3x3(g, "B", 1, False)
            g += 18
            main.append(("flush", "epiA"))
            main.append(Ins("", "wait_lds_writes"))
            main += self.stamp()
            pendingB = self.epi_layer("B", 1)
        # -- last layer's set B epilogue: exposed (the expand reads lo and hi from its first step) ----------------
        main.append(("call", lambda e: e.raw("\ts_barrier")))       # every wave is done reading hi
        main += pendingB
        main.append(Ins("", "wait_lds_writes"))
        main += self.stamp()
        # -- expand: four quarters of 64 output channels, sets A, B, A, B; a quarter's epilogue (residual add, x' -> HBM)
        # rides under the next quarter's K loop; the residual pieces alternate between the two X buffers -------------
        main += self.x_loads(0, 0, "r0")
        main += self.x_loads(1, 1, "r1")
        for qo in range(4):
            sset = "AB"[qo % 2]
            plan = [(c, [(sset, c == 0)]) for c in range(4)]
            main += self.phase1x1(g, plan)
            g += 4
            if qo > 0:
                main.append(("flush", f"xe{qo - 1}"))
                if qo + 1 < 4:
                    main += self.x_loads(qo + 1, (qo + 1) % 2, f"r{qo + 1}")
            main.append(("start", f"xe{qo}", self.epi_expand(qo, sset, qo % 2)))
        main.append(("flush", "xe3"))
        main += self.stamp()
        assert g == self.ngran, (g, self.ngran)
        # every wave is past the last K loop before the ring and the act buffer are reused
        main.append(("call", lambda e: e.raw("\ts_barrier")))
        return main

    # ---- whole kernel ---------------------------------------------------------------------------
    def kernel(self, name):
        e = self.e
        L = self.L
        self.kname = name
        e.raw(f"\t.globl\t{name}\n\t.p2align\t8\n\t.type\t{name},@function")
        e.label(name)
        # ---- prologue --------------------------------------------------------------------------
        e.raw(f"\ts_load_dwordx8 s[4:11], s[0:1], 0x0")       # x, wstream, prm, npos, nblk
        e.raw(f"\ts_load_dwordx4 s[12:15], s[0:1], 0x20")     # nwg, pad, stamps
        e.raw("\tv_and_b32 v1, 63, v0")                         # lane
        e.raw("\tv_lshrrev_b32 v2, 6, v0")                      # wave
        e.raw("\ts_nop 3")                                       # a VGPR write -> v_readfirstlane of it needs a wait state
        e.raw(f"\tv_readfirstlane_b32 s{S_WAVE}, v2")
        e.raw("\tv_and_b32 v3, 15, v1")                         # n
        e.raw("\tv_lshrrev_b32 v4, 4, v1")                      # q
        e.raw("\ts_nop 4")
        # V_DMAOFF = tid * 16
        e.raw(f"\tv_lshlrev_b32 v{V_DMAOFF + 100}, 4, v0")
        e.raw(f"\tv_lshlrev_b32 v{V_ARING + 100}, 4, v1")
        e.raw(f"\tv_add_u32 v{V_ARING + 100}, {RING0}, v{V_ARING + 100}")
        # rows: r0(b) = 32 * (wave + 4 b) + 2 n
        for b in range(3):
            e.raw(f"\ts_lshl_b32 s{S_TMP}, s{S_WAVE}, 5")
            e.raw(f"\ts_add_u32 s{S_TMP}, s{S_TMP}, {128 * b}")
            e.raw(f"\tv_lshl_add_u32 v5, v3, 1, s{S_TMP}")           # 2 n + 32 t
            # fragment base: r0 * SLOTB + q * 16   (rowshift of tap (ky = 0, kx = 0) is -PADTOP: cancels PADTOP)
            e.raw(f"\tv_mul_u32_u24 v6, {SLOTB}, v5")
            e.raw(f"\tv_lshl_add_u32 v{V_BROW + b + 100}, v4, 4, v6")
            # this lane's row after the pairing swap: r = r0 + (q & 1); piece chunk (q >> 1)
            e.raw("\tv_and_b32 v7, 1, v4")
            e.raw("\tv_add_u32 v7, v5, v7")                           # r
            e.raw("\tv_lshrrev_b32 v8, 1, v4")                        # q >> 1
            e.raw(f"\tv_add_u32 v9, {PADTOP}, v7")
            e.raw(f"\tv_mul_u32_u24 v9, {SLOTB}, v9")
            e.raw(f"\tv_lshl_add_u32 v{V_WROW + b + 100}, v8, 4, v9")
            # board point: y = r / 20, x = r % 20; on the board iff x < 19 and y < 19
            e.raw("\tv_mul_u32_u24 v10, 3277, v7")
            e.raw("\tv_lshrrev_b32 v10, 16, v10")                     # y = r * 3277 >> 16 (r < 1024)
            e.raw("\tv_mul_u32_u24 v11, 20, v10")
            e.raw("\tv_sub_u32 v11, v7, v11")                         # x
            e.raw("\tv_cmp_gt_u32 vcc, 19, v11")
            e.raw(f"\tv_cmp_gt_u32 s[{S_OK + 2 * b}:{S_OK + 2 * b + 1}], 19, v10")
            e.raw(f"\ts_and_b64 s[{S_OK + 2 * b}:{S_OK + 2 * b + 1}], s[{S_OK + 2 * b}:{S_OK + 2 * b + 1}], vcc")
            e.raw("\tv_mul_u32_u24 v12, 19, v10")
            e.raw("\tv_add_u32 v12, v12, v11")                        # loc
            e.raw(f"\tv_cndmask_b32 v12, 0, v12, s[{S_OK + 2 * b}:{S_OK + 2 * b + 1}]")
            e.raw("\tv_lshlrev_b32 v12, 4, v12")                      # loc * 16
            e.raw(f"\tv_mul_u32_u24 v13, {CBLK_BYTES}, v8")
            e.raw(f"\tv_add_u32 v{V_XOFF + b + 100}, v12, v13")
        e.raw(f"\tv_lshlrev_b32 v{V_QOFF + 100}, 4, v4")
        e.raw("\tv_lshrrev_b32 v8, 1, v4")
        e.raw(f"\tv_lshlrev_b32 v{V_POFF + 100}, 5, v8")
        # move the constants down into their registers (v100.. were scratch so that v0..v13 stayed free above)
        for r in list(range(V_DMAOFF, V_POFF + 1)):
            e.raw(f"\tv_mov_b32 v{r}, v{r + 100}")
        e.raw(f"\tv_mov_b32 v{V_C1}, {f32hex(-2.0 * LN2)}")
        e.raw(f"\tv_mov_b32 v{V_C2}, {f32hex(LN2)}")
        for i in range(4):
            e.raw(f"\tv_mov_b32 v{V_ZERO + i}, 0")
        e.raw(f"\ts_lshl_b32 s{S_RINGW}, s{S_WAVE}, 10")
        e.raw(f"\ts_add_u32 s{S_RINGW}, s{S_RINGW}, {RING0}")       # M0 base: the wave's 1 KiB piece of ring slot 0
        # zero the act buffer (the halo slots stay zero for the whole launch: epilogues write on-board rows only)
        e.raw("\tv_lshlrev_b32 v100, 4, v0")
        e.raw("\tv_mov_b32 v101, v100")
        nz = (ACT_BYTES + 4095) // 4096
        for i in range(nz):
            last = ACT_BYTES - i * 4096
            if last < 4096:
                e.raw(f"\tv_cmp_gt_u32 vcc, {last}, v101")
                e.raw("\ts_and_saveexec_b64 s[36:37], vcc")
            e.raw(f"\tds_write_b128 v100, v[{V_ZERO}:{V_ZERO + 3}]")
            if last < 4096:
                e.raw("\ts_mov_b64 exec, s[36:37]")
            else:
                e.raw("\tv_add_u32 v100, 4096, v100")
        e.raw("\ts_waitcnt lgkmcnt(0)")
        # flat loop state
        e.raw(f"\ts_mov_b32 s{S_POS}, s{S_WG}")
        e.raw(f"\ts_mov_b32 s{S_BLK}, 0")
        e.raw(f"\ts_cmp_ge_u32 s{S_POS}, s{S_NPOS}")
        e.raw(f"\ts_cbranch_scc0 .L{name}_go")
        self.long_jump(e, f".L{name}_end", False)
        e.label(f".L{name}_go")
        self.set_position(e)
        e.raw(f"\ts_mov_b64 s[{S_DMA}:{S_DMA + 1}], s[{S_WS}:{S_WS + 1}]")
        e.raw(f"\ts_mov_b64 s[{S_PB}:{S_PB + 1}], s[{S_PRM}:{S_PRM + 1}]")
        e.raw("\ts_barrier")
        if self.stop:
            # bisecting a fault on the GPU (tools/gpu_blockw_bisect.py): the pieces of the entry one by one, a marker to x[0]
            if self.stop.startswith("dump"):
                self.dump_and_end(e, tuple(self.stop.split("_")[1:]))
                e.label(f".L{name}_end")
                return
            pieces = {"prologue": [], "store": [], "xloads": self.x_loads(0, 0, "x0") + self.x_loads(1, 1, "x1"),
                      "prm": self.act_loads(0, 0), "dma": [self.dma_issue(g) for g in range(NSLOT)],
                      "entry": self.x_loads(0, 0, "x0") + self.x_loads(1, 1, "x1") + self.act_loads(0, 0) + [self.dma_issue(g) for g in range(NSLOT)]}
            for it in pieces[self.stop]:
                e.emit(it)
            e.raw("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
            if self.stop != "prologue":
                e.raw(f"\tv_mov_b32 v{TMP}, 0x3c003c00")
                e.raw(f"\tv_lshlrev_b32 v{TMP + 1}, 2, v0")
                e.raw(f"\tglobal_store_dword v{TMP + 1}, v{TMP}, s[{S_X}:{S_X + 1}]")
                e.raw("\ts_waitcnt vmcnt(0)")
            e.raw("\ts_endpgm")
            e.label(f".L{name}_end")
            return
        self.block_entry(e)
        e.label(f".L{name}_block")
        # loop-head state: the entry's operations (x quarters 0, 1; the block's first twelve granules) are the youngest
        # vector-memory operations; whatever is older (the previous block's stores) is never waited for by name
        head = [t for (_, t) in e.vm[-40:]]
        assert head == ["x0"] * 12 + ["x1"] * 12 + ["prm"] * 4 + [("g", g) for g in range(NSLOT)], head
        e.vm = list(e.vm[-40:])
        e.lgkm = []
        main = self.block_body()
        e.weave(main)
        # ---- advance (position, block) -------------------------------------------------------------
        if self.diag:
            self.write_stamps(e)
        e.raw(f"\ts_add_u32 s{S_BLK}, s{S_BLK}, 1")
        e.raw(f"\ts_cmp_lt_u32 s{S_BLK}, s{S_NBLK}")
        e.raw(f"\ts_cbranch_scc1 .L{name}_next")
        e.raw(f"\ts_mov_b32 s{S_BLK}, 0")
        e.raw(f"\ts_add_u32 s{S_POS}, s{S_POS}, s{S_NWG}")
        e.raw(f"\ts_cmp_ge_u32 s{S_POS}, s{S_NPOS}")
        e.raw(f"\ts_cbranch_scc1 .L{name}_end")
        self.set_position(e)
        e.label(f".L{name}_next")
        # weights and parameters of the block about to run; its first twelve granules
        e.raw(f"\ts_mul_i32 s{S_TMP}, s{S_BLK}, {self.ngran * GRAN}")
        e.raw(f"\ts_add_u32 s{S_DMA}, s{S_WS}, s{S_TMP}")
        e.raw(f"\ts_addc_u32 s{S_DMA + 1}, s{S_WS + 1}, 0")
        e.raw(f"\ts_mul_i32 s{S_TMP}, s{S_BLK}, {self.prm_floats * 4}")
        e.raw(f"\ts_add_u32 s{S_PB}, s{S_PRM}, s{S_TMP}")
        e.raw(f"\ts_addc_u32 s{S_PB + 1}, s{S_PRM + 1}, 0")
        self.block_entry(e)
        head = [t for (_, t) in e.vm[-40:]]
        assert head == ["x0"] * 12 + ["x1"] * 12 + ["prm"] * 4 + [("g", g) for g in range(NSLOT)], head
        self.long_jump(e, f".L{name}_block", True)
        e.label(f".L{name}_end")
        e.raw("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
        e.raw("\ts_endpgm")
        e.label(f".L{name}_fend")
        e.raw(f"\t.size\t{name}, .L{name}_fend-{name}")

    def block_entry(self, e):
        """x quarters 0, 1 of the block about to run are requested first, then its first twelve granules: the DMA issue
        covers part of the loads' latency, the rest of it is the one exposed wait per block"""
        e.raw("\ts_waitcnt lgkmcnt(0)")
        e.lgkm = []
        for it in self.x_loads(0, 0, "x0") + self.x_loads(1, 1, "x1") + self.act_loads(0, 0):
            e.emit(it)
        for g in range(NSLOT):
            e.emit(self.dma_issue(g))

    def long_jump(self, e, target, backward):
        """s_branch reaches +-128 KiB; a block body is larger"""
        self.njump = getattr(self, "njump", 0) + 1
        here = f".Lpc{self.njump}_{self.kname}"
        e.raw(f"\ts_getpc_b64 s[{S_TMP}:{S_TMP + 1}]")
        e.label(here)
        e.raw(f"\ts_add_u32 s{S_TMP}, s{S_TMP}, {target}-{here}")
        e.raw(f"\ts_addc_u32 s{S_TMP + 1}, s{S_TMP + 1}, {-1 if backward else 0}")
        e.raw(f"\ts_setpc_b64 s[{S_TMP}:{S_TMP + 1}]")

    def set_position(self, e):
        # S_XP = x + pos * POS_BYTES
        e.raw(f"\ts_mul_i32 s{S_TMP}, s{S_POS}, {POS_BYTES}")
        e.raw(f"\ts_mul_hi_u32 s{S_TMP + 1}, s{S_POS}, {POS_BYTES}")
        e.raw(f"\ts_add_u32 s{S_XP}, s{S_X}, s{S_TMP}")
        e.raw(f"\ts_addc_u32 s{S_XP + 1}, s{S_X + 1}, s{S_TMP + 1}")

    def write_stamps(self, e):
        # stamps of (first position of workgroups 0..7, every block): [wg][blk][wave][24] u64
        n = self.nstamp
        e.raw(f"\ts_cmp_lg_u64 s[{S_STAMP}:{S_STAMP + 1}], 0")
        e.raw(f"\ts_cbranch_scc0 .Lnostamp_{self.kname}")
        e.raw(f"\ts_cmp_lt_u32 s{S_WG}, 8")
        e.raw(f"\ts_cbranch_scc0 .Lnostamp_{self.kname}")
        e.raw(f"\ts_cmp_eq_u32 s{S_POS}, s{S_WG}")
        e.raw(f"\ts_cbranch_scc0 .Lnostamp_{self.kname}")
        # offset = ((wg * 16 + blk) * 4 + wave) * 24 * 8
        e.raw(f"\ts_lshl_b32 s{S_TMP}, s{S_WG}, 4")
        e.raw(f"\ts_add_u32 s{S_TMP}, s{S_TMP}, s{S_BLK}")
        e.raw(f"\ts_lshl_b32 s{S_TMP}, s{S_TMP}, 2")
        e.raw(f"\ts_add_u32 s{S_TMP}, s{S_TMP}, s{S_WAVE}")
        e.raw(f"\ts_mul_i32 s{S_TMP}, s{S_TMP}, 192")
        e.raw(f"\ts_add_u32 s{S_TMP + 2}, s{S_STAMP}, s{S_TMP}")
        e.raw(f"\ts_addc_u32 s{S_TMP + 3}, s{S_STAMP + 1}, 0")
        e.raw(f"\ts_getreg_b32 s{S_T0 + 46}, hwreg(HW_REG_LDS_ALLOC)")
        e.raw(f"\ts_mov_b32 s{S_T0 + 47}, 0")
        e.raw(f"\tv_mov_b32 v{TMP}, 0")
        for k in list(range(n)) + [23]:
            e.raw(f"\tv_mov_b32 v{TMP + 2}, s{S_T0 + 2 * k}")
            e.raw(f"\tv_mov_b32 v{TMP + 3}, s{S_T0 + 2 * k + 1}")
            e.raw(f"\tglobal_store_dwordx2 v{TMP}, v[{TMP + 2}:{TMP + 3}], s[{S_TMP + 2}:{S_TMP + 3}] offset:{8 * k}")
        e.raw("\ts_waitcnt vmcnt(0)")
        e.label(f".Lnostamp_{self.kname}")


def descriptor(name, lds):
    return f"""
	.section	.rodata,"a",@progbits
	.p2align	6, 0x0
	.amdhsa_kernel {name}
		.amdhsa_group_segment_fixed_size {lds}
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size 64
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 512
		.amdhsa_next_free_sgpr 96
		.amdhsa_accum_offset 256
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
	.end_amdhsa_kernel
	.text
"""


def metadata(names, lds):
    ks = ""
    for n in names:
        ks += f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           64
        .value_kind:     by_value
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: 64
    .max_flat_workgroup_size: 256
    .name:           {n}
    .private_segment_fixed_size: 0
    .sgpr_count:     96
    .sgpr_spill_count: 0
    .symbol:         {n}.kd
    .uniform_work_group_size: 1
    .vgpr_count:     512
    .vgpr_spill_count: 0
    .wavefront_size: 64
"""
    return f"""	.amdgpu_metadata
---
amdhsa.kernels:
{ks}amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
	.end_amdgpu_metadata
"""


def main():
    out = sys.argv[1]
    variants = [(L, d, None) for L in (1, 2, 3) for d in (False, True)]
    if len(sys.argv) > 2:
        variants = [(int(sys.argv[2]), False, int(sys.argv[3]) if len(sys.argv) > 3 else None)]
    text = '\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.text\n'
    names = []
    for L, diag, dump in variants:
        name = f"k_blockw_L{L}" + ("_diag" if diag else "")
        g = BlockGen(L, diag, dump)
        g.kernel(name)
        text += g.e.text() + descriptor(name, LDS_BYTES)
        names.append(name)
        print(name, g.e.stats, "lines", len(g.e.lines), file=sys.stderr)
    text += metadata(names, LDS_BYTES)
    with open(out, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
