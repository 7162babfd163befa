"""gfx950 assembly emitter for the hand-scheduled trunk kernel (blockw_gen.py).

What it does that an assembler does not:
  * counted s_waitcnt: every LDS / vector-memory operation is tracked in issue order, and a wait with the exact
    count (ops issued after the awaited one may stay in flight) is emitted right before the first instruction that
    reads or overwrites a register such an operation still has to deliver (vmcnt and lgkmcnt complete in issue
    order for loads, stores, LDS-DMA and LDS accesses: MI355X_MICROARCH.md, cycle constants, last paragraph);
  * software wait states: the hazards the hardware does not interlock (MFMA result -> any other reader, transcendental
    -> consumer, VALU -> permlane swap, wide store data -> overwrite, ...) are padded with s_nop from a conservative
    table, measured in issued instructions;
  * weaving: a main stream (the MFMA K loops with their LDS fragment reads and ring syncs) is interleaved with filler
    streams (the BN + mish epilogues, activations, residual adds) at a fixed issue budget per MFMA, which is how a
    single wave per SIMD keeps both the matrix pipe and the vector ALU busy (MI355X_MICROARCH.md: an MFMA
    16x16x32 holds the vector issue for 8 of its 16 cycles; single-issue fillers up to the rest are hidden).
"""
from collections import deque


def V(i, n=1):
    return ("v", i, n)


def A(i, n=1):
    return ("a", i, n)


def S(i, n=1):
    return ("s", i, n)


def rtxt(r):
    """register operand text; non-register operands (ints, floats, strings) pass through"""
    if isinstance(r, tuple):
        k, i, n = r
        return f"{k}{i}" if n == 1 else f"{k}[{i}:{i + n - 1}]"
    return str(r)


def regs(*ops):
    out = set()
    for r in ops:
        if isinstance(r, tuple):
            k, i, n = r
            for j in range(n):
                out.add((k, i + j))
    return out


class Ins:
    __slots__ = ("text", "kind", "dst", "src", "cost", "note", "inflight_dst")

    def __init__(self, text, kind, dst=(), src=(), cost=4, note=None):
        self.text = text
        self.kind = kind        # mfma valu trans perm salu ds_r ds_w vm_ld vm_st dma smem misc
        self.dst = regs(*dst)
        self.src = regs(*src)
        self.cost = cost        # issue cycles, for the weaver's budget
        self.note = note


class Bundle:
    """instructions that must stay contiguous (an EXEC-masked store between its mask and its restore)"""

    def __init__(self, items):
        self.items = items
        self.cost = sum(i.cost for i in items)


# minimum distance, in issued instructions, between a producer of kind P and a dependent consumer of kind C
# (conservative; the producer's destination is read or overwritten by the consumer)
def _need_gap(p, c, raw):
    if p.kind == "mfma":
        if c.kind == "mfma":
            return 0              # accumulate chain on the same registers: interlocked
        return 20 if p.note == "mfma32" else 16   # 16- / 8-pass XDL result -> any other reader / writer
    if p.kind == "trans":
        return 2
    if p.kind in ("valu", "perm"):
        if c.kind == "perm" or p.kind == "perm":
            return 3
        if c.kind == "mfma":
            return 3
        if c.kind in ("vm_ld", "vm_st", "dma", "ds_r", "ds_w") and not raw:
            return 0
        return 2 if p.note == "dstsel" else 0
    if p.kind == "salu":
        if c.kind in ("vm_ld", "vm_st", "dma"):
            return 1
        return 0
    if p.kind == "rfl":           # v_readfirstlane: VALU writes an SGPR
        return 5
    return 0


class Emitter:
    def __init__(self):
        self.lines = []
        self.hist = deque(maxlen=24)      # (index, Ins) of recent real instructions
        self.count = 0                    # issued instructions (s_nop N counts N + 1)
        self.lgkm = []                    # outstanding LDS ops in issue order: (dst regs, kind)
        self.vm = []                      # outstanding vector-memory ops in issue order: (dst regs, tag)
        self.stats = {"mfma": 0, "nop": 0, "wait_lgkm": 0, "wait_vm": 0}
        self.store_guard = deque(maxlen=4)   # (index, data regs) of recent wide stores

    # ---- raw output ----------------------------------------------------------------------
    def raw(self, text):
        self.lines.append(text)

    def comment(self, text):
        self.lines.append(f"\t; {text}")

    def label(self, name):
        self.lines.append(f"{name}:")

    def _nop(self, n):
        while n > 0:
            k = min(n, 16)
            self.lines.append(f"\ts_nop {k - 1}")
            self.count += k
            self.stats["nop"] += k
            n -= k

    # ---- waits -----------------------------------------------------------------------------
    def prewait(self, regs_):
        """one wait for everything in `regs_` that is still in flight (a K step's fragments at its start: each
        s_waitcnt costs an issue slot, so one per step instead of one per fragment)"""
        self.wait_regs(regs(*regs_))

    def _wait_lgkm(self, n):
        self.lines.append(f"\ts_waitcnt lgkmcnt({n})")
        self.stats["wait_lgkm"] += 1
        self.count += 1
        del self.lgkm[: len(self.lgkm) - n]

    def _wait_vm(self, n):
        self.lines.append(f"\ts_waitcnt vmcnt({n})")
        self.stats["wait_vm"] += 1
        self.count += 1
        del self.vm[: len(self.vm) - n]

    def wait_regs(self, touched):
        """waits until no outstanding LDS / VMEM operation still writes any of `touched`"""
        for q, waiter, cap in ((self.lgkm, self._wait_lgkm, 15), (self.vm, self._wait_vm, 63)):
            last = -1
            for i, (dst, _) in enumerate(q):
                if dst & touched:
                    last = i
            if last >= 0:
                waiter(min(len(q) - 1 - last, cap))

    def wait_lgkm_all(self):
        if self.lgkm:
            self._wait_lgkm(0)

    def wait_lds_writes(self):
        """every LDS store issued so far has completed (before a barrier that publishes it)"""
        last = -1
        for i, (_, kind) in enumerate(self.lgkm):
            if kind == "ds_w":
                last = i
        if last >= 0:
            self._wait_lgkm(min(len(self.lgkm) - 1 - last, 15))

    def wait_vm_tags(self, tags):
        """every vector-memory operation whose tag is in `tags` has completed"""
        last = -1
        for i, (_, tag) in enumerate(self.vm):
            if tag in tags:
                last = i
        if last >= 0:
            self._wait_vm(min(len(self.vm) - 1 - last, 63))

    def wait_vm_all(self):
        if self.vm:
            self._wait_vm(0)

    # ---- one instruction -----------------------------------------------------------------
    def emit(self, ins):
        if isinstance(ins, Bundle):
            for i in ins.items:
                self.emit(i)
            return
        if ins.kind == "wait_lds_writes":
            self.wait_lds_writes()
            return
        touched = ins.dst | ins.src
        self.wait_regs(touched)
        # software wait states
        need = 0
        for idx, p in self.hist:
            raw = bool(p.dst & ins.src)
            waw = bool(p.dst & ins.dst)
            if raw or waw:
                g = _need_gap(p, ins, raw)
                need = max(need, g - (self.count - idx - 1))
        if ins.dst and ins.kind not in ("mfma",):
            for idx, data in self.store_guard:
                if data & ins.dst:
                    need = max(need, 2 - (self.count - idx - 1))
        if need > 0:
            self._nop(need)
        self.lines.append("\t" + ins.text)
        if ins.kind == "mfma":
            self.stats["mfma"] += 1
        self.hist.append((self.count, ins))
        if ins.kind in ("vm_st", "ds_w"):
            self.store_guard.append((self.count, set(ins.src)))
        self.count += 1
        if ins.kind in ("ds_r", "ds_w", "smem"):
            self.lgkm.append((set(ins.dst), ins.kind))
        elif ins.kind in ("vm_ld", "vm_st", "dma"):
            self.vm.append((set(ins.dst), ins.note))

    # ---- weaving ----------------------------------------------------------------------------
    def weave(self, main, budget=24, mfma_cost=8):
        """main: list of items; an item is an Ins, a Bundle, or a control tuple:
             ("start", name, [filler items])   the filler stream `name` may be issued from here on
             ("flush", name)                   everything left of stream `name` is issued here
             ("call", fn)                      fn(self) runs at this point (barriers, explicit waits)
           After every MFMA the pending filler streams are drained, oldest first, while the gap's issue budget lasts."""
        pending = []            # [name, deque(items)]
        credit = 0
        for it in main:
            if isinstance(it, tuple):
                if it[0] == "start":
                    pending.append([it[1], deque(it[2])])
                elif it[0] == "flush":
                    for p in pending:
                        if p[0] == it[1]:
                            while p[1]:
                                self.emit(p[1].popleft())
                    pending = [p for p in pending if p[0] != it[1]]
                elif it[0] == "call":
                    it[1](self)
                continue
            self.emit(it)
            if isinstance(it, Ins) and it.kind == "mfma":
                # what the main stream spent since the last MFMA comes out of this gap, and a bounded part of an older debt
                # (an exposed stretch of main-stream VALU work must not starve the fillers of the K loop behind it)
                credit = max(min(credit, 0), -2 * budget) + budget
                while pending and credit > 0:
                    name, q = pending[0]
                    if not q:
                        pending.pop(0)
                        continue
                    nxt = q[0]
                    if nxt.cost > credit + 4 and credit < budget:
                        break
                    self.emit(q.popleft())
                    credit -= nxt.cost
            elif isinstance(it, Ins):
                credit -= it.cost
            elif isinstance(it, Bundle):
                credit -= it.cost
        for name, q in pending:
            if q:
                raise RuntimeError(f"filler stream {name} was never flushed ({len(q)} items left)")

    def text(self):
        return "\n".join(self.lines) + "\n"


# ---- instruction constructors -------------------------------------------------------------
def mfma(acc_out, a, b, acc_in):
    c = rtxt(acc_in) if isinstance(acc_in, tuple) else "0"
    src = [a, b] + ([acc_in] if isinstance(acc_in, tuple) else [])
    return Ins(f"v_mfma_f32_16x16x32_f16 {rtxt(acc_out)}, {rtxt(a)}, {rtxt(b)}, {c}", "mfma", [acc_out], src, 8)


def mfma32(acc_out, a, b, acc_in):
    """v_mfma_f32_32x32x16_f16: 16 passes (32 cycles on the matrix pipe), 8 of them hold the wave's issue"""
    c = rtxt(acc_in) if isinstance(acc_in, tuple) else "0"
    src = [a, b] + ([acc_in] if isinstance(acc_in, tuple) else [])
    return Ins(f"v_mfma_f32_32x32x16_f16 {rtxt(acc_out)}, {rtxt(a)}, {rtxt(b)}, {c}", "mfma", [acc_out], src, 8, "mfma32")


def permswap32(x, y):
    return Ins(f"v_permlane32_swap_b32 {rtxt(x)}, {rtxt(y)}", "perm", [x, y], [x, y], 4)


def cvt_pk(dst, a, b):
    return Ins(f"v_cvt_pk_f16_f32 {rtxt(dst)}, {rtxt(a)}, {rtxt(b)}", "valu", [dst], [a, b], 4)


def ds_read128(dst, addr, off):
    assert 0 <= off < 65536, off
    return Ins(f"ds_read_b128 {rtxt(dst)}, {rtxt(addr)} offset:{off}", "ds_r", [dst], [addr], 4)


def ds_write128(addr, data, off):
    assert 0 <= off < 65536, off
    return Ins(f"ds_write_b128 {rtxt(addr)}, {rtxt(data)} offset:{off}", "ds_w", [], [addr, data], 8)


def gload128(dst, voff, sbase, off=0, tag=None):
    assert -4096 <= off < 4096, off
    o = f" offset:{off}" if off else ""
    return Ins(f"global_load_dwordx4 {rtxt(dst)}, {rtxt(voff)}, {rtxt(sbase)}{o}", "vm_ld", [dst], [voff, sbase], 4, tag)


def gstore128(voff, data, sbase, off=0, tag=None):
    assert -4096 <= off < 4096, off
    o = f" offset:{off}" if off else ""
    return Ins(f"global_store_dwordx4 {rtxt(voff)}, {rtxt(data)}, {rtxt(sbase)}{o}", "vm_st", [], [voff, data, sbase], 4, tag)


def valu(op, dst, *src, cost=4, mods="", note=None):
    ops = ", ".join(rtxt(x) for x in (dst,) + src)
    return Ins(f"{op} {ops}{mods}", "valu", [dst], list(src), cost, note)


def trans(op, dst, src):
    return Ins(f"{op} {rtxt(dst)}, {rtxt(src)}", "trans", [dst], [src], 8)


def accread(dst, a):
    return Ins(f"v_accvgpr_read_b32 {rtxt(dst)}, {rtxt(a)}", "valu", [dst], [a], 4)


def permswap16(x, y):
    return Ins(f"v_permlane16_swap_b32 {rtxt(x)}, {rtxt(y)}", "perm", [x, y], [x, y], 4)


def salu(op, dst, *src, cost=4):
    ops = ", ".join(rtxt(x) for x in (dst,) + src)
    return Ins(f"{op} {ops}", "salu", [dst] if isinstance(dst, tuple) else [], list(src), cost)


def misc(text, cost=4):
    return Ins(text, "misc", [], [], cost)
