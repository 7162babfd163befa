"""Functional emulator of the gfx950 instruction subset blockw_gen.py emits (test infrastructure).

It executes the generated assembly TEXT for one workgroup (four waves of 64 lanes) in program order: registers, EXEC,
LDS and global memory are modelled; timing, s_waitcnt counters and wait states are not (the emitter's own bookkeeping
covers those; the GPU parity tests cover what this cannot see).  Waves run one after the other from barrier to barrier,
which is exact for this kernel: waves exchange data through LDS only across s_barrier.
Used by tests/test_blockw_asm_cpu.py to check the kernel against a numpy restatement of the bottleneck block without a
GPU, and to localise a wrong value to the instruction that produced it.
"""
import re

import numpy as np

NLANE = 64


def _f32(u):
    return u.view(np.float32)


def _u32(f):
    return np.asarray(f, np.float32).view(np.uint32)


class Mem:
    """global memory: named buffers at fixed fake addresses"""

    def __init__(self):
        self.bufs = []      # (base, bytearray-like np.uint8 array)
        self.next = 0x100000

    def add(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.next += (len(a) + 0xFFFF) // 0x10000 * 0x10000 + 0x10000
        self.bufs.append((base, a))
        return base

    def find(self, addr, n):
        for base, a in self.bufs:
            if base <= addr and addr + n <= base + len(a):
                return a, addr - base
        raise RuntimeError(f"global access out of bounds: {addr:#x} + {n}")

    def read(self, addr, n):
        a, o = self.find(addr, n)
        return a[o:o + n]

    def write(self, addr, data):
        a, o = self.find(addr, len(data))
        a[o:o + len(data)] = data

    def array(self, base, dtype, count):
        a, o = self.find(base, count * np.dtype(dtype).itemsize)
        return a[o:o + count * np.dtype(dtype).itemsize].view(dtype)


class Wave:
    def __init__(self, wid):
        self.v = np.zeros((512, NLANE), np.uint32)     # 0..255 VGPR, 256..511 AGPR
        self.s = np.zeros(128, np.uint32)
        self.vcc = np.uint64(0)
        self.exec = np.uint64(0xFFFFFFFFFFFFFFFF)
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.done = False
        self.wid = wid
        self.count = 0


_REG = re.compile(r"^([vas])(?:(\d+)|\[(\d+):(\d+)\])$")


def parse_operand(t):
    t = t.strip()
    m = _REG.match(t)
    if m:
        k = m.group(1)
        if m.group(2) is not None:
            return (k, int(m.group(2)), 1)
        return (k, int(m.group(3)), int(m.group(4)) - int(m.group(3)) + 1)
    if t in ("vcc", "exec", "m0", "off", "scc"):
        return (t,)
    if t.startswith("0x"):
        return ("imm", int(t, 16))
    try:
        if "." in t:
            return ("fimm", float(t))
        return ("imm", int(t))
    except ValueError:
        return ("sym", t)


class Program:
    def __init__(self, text, kernel):
        self.ins = []
        self.labels = {}
        on = False
        for line in text.splitlines():
            line = line.split(";")[0].rstrip()
            if not line.strip():
                continue
            if not line.startswith("\t") and line.endswith(":"):
                name = line[:-1]
                if name == kernel:
                    on = True
                if on:
                    self.labels[name] = len(self.ins)
                continue
            if not on:
                continue
            st = line.strip()
            if st.startswith("."):
                if st.startswith(".size"):
                    break
                continue
            parts = st.split(None, 1)
            op = parts[0]
            rest = parts[1] if len(parts) > 1 else ""
            mods = {}
            for m in re.finditer(r"(op_sel_hi|op_sel):\[([\d,]+)\]", rest):
                mods[m.group(1)] = [int(x) for x in m.group(2).split(",")]
            rest = re.sub(r"(op_sel_hi|op_sel):\[[\d,]+\]", "", rest)
            m = re.search(r"offset:(-?\d+)", rest)
            if m:
                mods["offset"] = int(m.group(1))
                rest = rest.replace(m.group(0), "")
            if op == "s_waitcnt":
                self.ins.append((op, [], mods))
                continue
            if op.startswith("s_getreg"):
                self.ins.append(("s_nop", [], mods))
                continue
            ops = []
            depth = 0
            cur = ""
            for ch in rest:
                if ch == "[":
                    depth += 1
                if ch == "]":
                    depth -= 1
                if ch == "," and depth == 0:
                    ops.append(cur)
                    cur = ""
                else:
                    cur += ch
            if cur.strip():
                ops.append(cur)
            self.ins.append((op, [parse_operand(o) for o in ops if o.strip()], mods))


class Sim:
    def __init__(self, text, kernel, mem, kernarg_addr, wg_id, lds_bytes=163840):
        self.p = Program(text, kernel)
        self.mem = mem
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.waves = [Wave(w) for w in range(4)]
        for w in self.waves:
            w.s[0] = kernarg_addr & 0xFFFFFFFF
            w.s[1] = kernarg_addr >> 32
            w.s[2] = wg_id
            w.v[0] = np.arange(NLANE, dtype=np.uint32) + 64 * w.wid
        self.trace = None

    # ---- operand access --------------------------------------------------------------------
    def rd32(self, w, o, j=0):
        """32-bit value(s) of operand o (register j of a range): per-lane array or scalar broadcast"""
        k = o[0]
        if k == "v":
            return w.v[o[1] + j]
        if k == "a":
            return w.v[256 + o[1] + j]
        if k == "s":
            return np.uint32(w.s[o[1] + j])
        if k == "imm":
            return np.uint32(o[1] & 0xFFFFFFFF)
        if k == "fimm":
            return _u32(np.float32(o[1]))[()]
        if k == "vcc":
            return np.uint32((int(w.vcc) >> (32 * j)) & 0xFFFFFFFF)
        if k == "exec":
            return np.uint32((int(w.exec) >> (32 * j)) & 0xFFFFFFFF)
        if k == "m0":
            return np.uint32(w.m0)
        raise RuntimeError(f"operand {o}")

    def rdf(self, w, o):
        """float operand: integer inline constants of float instructions are not used by the generator except 0"""
        if o[0] == "imm":
            v = o[1]
            if v == 0:
                return np.float32(0.0)
            return _f32(np.array([v & 0xFFFFFFFF], np.uint32))[0]
        if o[0] == "fimm":
            return np.float32(o[1])
        x = self.rd32(w, o)
        return _f32(np.asarray(x, np.uint32).reshape(-1)) if isinstance(x, np.ndarray) else _f32(np.array([x], np.uint32))[0]

    def lanes(self, w):
        return ((int(w.exec) >> np.arange(NLANE, dtype=np.uint64).astype(object)) & 1).astype(bool) if False else \
            np.array([(int(w.exec) >> i) & 1 for i in range(NLANE)], bool)

    def wr32(self, w, o, val, j=0, mask=None):
        k = o[0]
        if k in ("v", "a"):
            idx = o[1] + j + (256 if k == "a" else 0)
            val = np.broadcast_to(np.asarray(val, np.uint32), (NLANE,))
            if mask is None:
                mask = self.lanes(w)
            w.v[idx] = np.where(mask, val, w.v[idx])
        elif k == "s":
            w.s[o[1] + j] = np.uint32(int(val) & 0xFFFFFFFF)
        elif k == "m0":
            w.m0 = int(val) & 0xFFFFFFFF
        elif k == "vcc":
            w.vcc = np.uint64(int(val))
        elif k == "exec":
            w.exec = np.uint64(int(val))
        else:
            raise RuntimeError(f"dst {o}")

    def s64(self, w, o):
        if o[0] == "s":
            return int(w.s[o[1]]) | (int(w.s[o[1] + 1]) << 32)
        if o[0] == "vcc":
            return int(w.vcc)
        if o[0] == "exec":
            return int(w.exec)
        if o[0] == "imm":
            return o[1] & 0xFFFFFFFFFFFFFFFF if o[1] >= 0 else (o[1] + (1 << 64))
        raise RuntimeError(f"s64 {o}")

    def w64(self, w, o, val):
        val &= 0xFFFFFFFFFFFFFFFF
        if o[0] == "s":
            w.s[o[1]] = np.uint32(val & 0xFFFFFFFF)
            w.s[o[1] + 1] = np.uint32(val >> 32)
        elif o[0] == "vcc":
            w.vcc = np.uint64(val)
        elif o[0] == "exec":
            w.exec = np.uint64(val)
        else:
            raise RuntimeError(f"w64 {o}")

    # ---- execution --------------------------------------------------------------------------
    def run(self, max_steps=10**9):
        """runs all waves to completion; returns the number of barriers passed"""
        nbar = 0
        while not all(w.done for w in self.waves):
            states = []
            for w in self.waves:
                if not w.done:
                    states.append(self.run_wave(w, max_steps))
            if all(s == "barrier" for s in states):
                nbar += 1
            elif any(s == "barrier" for s in states):
                raise RuntimeError(f"waves disagree at a barrier: {states}")
        return nbar

    def run_wave(self, w, max_steps):
        ins = self.p.ins
        n = 0
        while True:
            op, o, mods = ins[w.pc]
            w.pc += 1
            w.count += 1
            n += 1
            if n > max_steps:
                raise RuntimeError("step limit")
            r = self.step(w, op, o, mods)
            if r is not None:
                return r

    def step(self, w, op, o, mods):
        if op in ("s_waitcnt", "s_nop", "s_setprio", "s_sleep"):
            return None
        if op == "s_barrier":
            return "barrier"
        if op == "s_endpgm":
            w.done = True
            return "done"
        f = getattr(self, "op_" + op, None)
        if f is None:
            raise RuntimeError(f"unimplemented instruction {op}")
        return f(w, o, mods)

    # ---- SALU -------------------------------------------------------------------------------
    def op_s_load_dwordx8(self, w, o, mods, n=8):
        base = self.s64(w, o[1]) + (o[2][1] if len(o) > 2 else 0)
        data = self.mem.read(base, 4 * n).view(np.uint32)
        for j in range(n):
            w.s[o[0][1] + j] = data[j]

    def op_s_load_dwordx4(self, w, o, mods):
        self.op_s_load_dwordx8(w, o, mods, 4)

    def op_s_load_dwordx2(self, w, o, mods):
        self.op_s_load_dwordx8(w, o, mods, 2)

    def op_s_mov_b32(self, w, o, mods):
        self.wr32(w, o[0], self.rd32(w, o[1]))

    def op_s_mov_b64(self, w, o, mods):
        self.w64(w, o[0], self.s64(w, o[1]))

    def op_s_add_u32(self, w, o, mods):
        if o[2][0] == "sym":     # long jump: label difference
            a, b = o[2][1].split("-")
            val = (self.p.labels[a] - self.p.labels[b]) & 0xFFFFFFFF
            w.s[o[0][1]] = np.uint32(val)      # holds the target instruction index delta; s_setpc uses it
            w.jump_delta = self.p.labels[a]
            w.scc = 0
            return
        r = int(self.rd32(w, o[1])) + int(self.rd32(w, o[2]))
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self.wr32(w, o[0], r & 0xFFFFFFFF)

    def op_s_addc_u32(self, w, o, mods):
        r = int(self.rd32(w, o[1])) + int(self.rd32(w, o[2])) + w.scc
        w.scc = 1 if r > 0xFFFFFFFF else 0
        self.wr32(w, o[0], r & 0xFFFFFFFF)

    def op_s_getpc_b64(self, w, o, mods):
        self.w64(w, o[0], 0)

    def op_s_setpc_b64(self, w, o, mods):
        w.pc = w.jump_delta

    def op_s_lshl_b32(self, w, o, mods):
        r = (int(self.rd32(w, o[1])) << (int(self.rd32(w, o[2])) & 31)) & 0xFFFFFFFF
        w.scc = 1 if r else 0
        self.wr32(w, o[0], r)

    def op_s_mul_i32(self, w, o, mods):
        self.wr32(w, o[0], (int(self.rd32(w, o[1])) * int(self.rd32(w, o[2]))) & 0xFFFFFFFF)

    def op_s_mul_hi_u32(self, w, o, mods):
        self.wr32(w, o[0], (int(self.rd32(w, o[1])) * int(self.rd32(w, o[2]))) >> 32)

    def _cmp(self, w, o, fn, bits=32):
        if bits == 64:
            w.scc = 1 if fn(self.s64(w, o[0]), self.s64(w, o[1])) else 0
        else:
            w.scc = 1 if fn(int(self.rd32(w, o[0])), int(self.rd32(w, o[1]))) else 0

    def op_s_cmp_ge_u32(self, w, o, mods):
        self._cmp(w, o, lambda a, b: a >= b)

    def op_s_cmp_lt_u32(self, w, o, mods):
        self._cmp(w, o, lambda a, b: a < b)

    def op_s_cmp_eq_u32(self, w, o, mods):
        self._cmp(w, o, lambda a, b: a == b)

    def op_s_cmp_lg_u64(self, w, o, mods):
        self._cmp(w, o, lambda a, b: a != b, 64)

    def op_s_cbranch_scc1(self, w, o, mods):
        if w.scc:
            w.pc = self.p.labels[o[0][1]]

    def op_s_cbranch_scc0(self, w, o, mods):
        if not w.scc:
            w.pc = self.p.labels[o[0][1]]

    def op_s_branch(self, w, o, mods):
        w.pc = self.p.labels[o[0][1]]

    def op_s_and_b64(self, w, o, mods):
        r = self.s64(w, o[1]) & self.s64(w, o[2])
        w.scc = 1 if r else 0
        self.w64(w, o[0], r)

    def op_s_and_saveexec_b64(self, w, o, mods):
        old = int(w.exec)
        self.w64(w, o[0], old)
        w.exec = np.uint64(old & self.s64(w, o[1]))

    def op_s_memtime(self, w, o, mods):
        self.w64(w, o[0], w.count)

    # ---- VALU integer -------------------------------------------------------------------------
    def _v(self, w, o):
        return np.broadcast_to(np.asarray(self.rd32(w, o), np.uint32), (NLANE,)).astype(np.uint64)

    def op_v_and_b32(self, w, o, mods):
        self.wr32(w, o[0], (self._v(w, o[1]) & self._v(w, o[2])).astype(np.uint32))

    def op_v_lshrrev_b32(self, w, o, mods):
        self.wr32(w, o[0], (self._v(w, o[2]) >> (self._v(w, o[1]) & 31)).astype(np.uint32))

    def op_v_lshlrev_b32(self, w, o, mods):
        self.wr32(w, o[0], ((self._v(w, o[2]) << (self._v(w, o[1]) & 31)) & 0xFFFFFFFF).astype(np.uint32))

    def op_v_add_u32(self, w, o, mods):
        self.wr32(w, o[0], ((self._v(w, o[1]) + self._v(w, o[2])) & 0xFFFFFFFF).astype(np.uint32))

    def op_v_sub_u32(self, w, o, mods):
        self.wr32(w, o[0], ((self._v(w, o[1]) - self._v(w, o[2])) & 0xFFFFFFFF).astype(np.uint32))

    def op_v_lshl_add_u32(self, w, o, mods):
        self.wr32(w, o[0], (((self._v(w, o[1]) << (self._v(w, o[2]) & 31)) + self._v(w, o[3])) & 0xFFFFFFFF).astype(np.uint32))

    def op_v_mul_u32_u24(self, w, o, mods):
        self.wr32(w, o[0], (((self._v(w, o[1]) & 0xFFFFFF) * (self._v(w, o[2]) & 0xFFFFFF)) & 0xFFFFFFFF).astype(np.uint32))

    def op_v_cmp_gt_u32(self, w, o, mods):
        r = self._v(w, o[1]) > self._v(w, o[2])
        bits = 0
        act = self.lanes(w)
        for i in range(NLANE):
            if r[i] and act[i]:
                bits |= 1 << i
        self.w64(w, o[0], bits)

    def op_v_cndmask_b32(self, w, o, mods):
        sel = self.s64(w, o[3])
        m = np.array([(sel >> i) & 1 for i in range(NLANE)], bool)
        self.wr32(w, o[0], np.where(m, self._v(w, o[2]), self._v(w, o[1])).astype(np.uint32))

    def op_v_mov_b32(self, w, o, mods):
        self.wr32(w, o[0], self._v(w, o[1]).astype(np.uint32))

    def op_v_readfirstlane_b32(self, w, o, mods):
        act = self.lanes(w)
        first = int(np.argmax(act))
        self.wr32(w, o[0], self._v(w, o[1])[first])

    def op_v_writelane_b32(self, w, o, mods):
        idx = o[0][1]
        w.v[idx, int(self.rd32(w, o[2])) & 63] = np.uint32(int(self.rd32(w, o[1])))

    def op_v_accvgpr_read_b32(self, w, o, mods):
        self.wr32(w, o[0], self._v(w, o[1]).astype(np.uint32))

    def op_v_permlane16_swap_b32(self, w, o, mods):
        x = self._v(w, o[0]).astype(np.uint32).copy()
        y = self._v(w, o[1]).astype(np.uint32).copy()
        nx, ny = x.copy(), y.copy()
        for h in (0, 32):
            nx[h + 16:h + 32] = y[h:h + 16]
            ny[h:h + 16] = x[h + 16:h + 32]
        full = np.ones(NLANE, bool)
        self.wr32(w, o[0], nx, mask=full)
        self.wr32(w, o[1], ny, mask=full)

    # ---- VALU float ----------------------------------------------------------------------------
    def _fl(self, w, o):
        if o[0] in ("imm", "fimm"):
            return np.full(NLANE, self.rdf(w, o), np.float32)
        return _f32(self._v(w, o).astype(np.uint32))

    def op_v_add_f32(self, w, o, mods):
        self.wr32(w, o[0], _u32(self._fl(w, o[1]) + self._fl(w, o[2])))

    def op_v_fma_f32(self, w, o, mods):
        a, b, c = (self._fl(w, x).astype(np.float64) for x in o[1:4])
        self.wr32(w, o[0], _u32((a * b + c).astype(np.float32)))

    def op_v_exp_f32(self, w, o, mods):
        with np.errstate(over="ignore"):
            self.wr32(w, o[0], _u32(np.exp2(self._fl(w, o[1]).astype(np.float64)).astype(np.float32)))

    def op_v_rcp_f32(self, w, o, mods):
        with np.errstate(divide="ignore"):
            self.wr32(w, o[0], _u32((1.0 / self._fl(w, o[1]).astype(np.float64)).astype(np.float32)))

    def _mixsrc(self, w, o, idx, mods):
        sel = mods.get("op_sel", [0, 0, 0])[idx]
        hi = mods.get("op_sel_hi", [0, 0, 0])[idx]
        if o[0] in ("imm", "fimm"):
            return np.full(NLANE, self.rdf(w, o), np.float64)
        u = self._v(w, o).astype(np.uint32)
        if hi:      # f16 source, half chosen by op_sel
            h = ((u >> 16) if sel else (u & 0xFFFF)).astype(np.uint16)
            return h.view(np.float16).astype(np.float64)
        return _f32(u).astype(np.float64)

    def _mix(self, w, o, mods):
        a, b, c = (self._mixsrc(w, o[1 + i], i, mods) for i in range(3))
        return (a * b + c).astype(np.float32)

    def op_v_fma_mix_f32(self, w, o, mods):
        self.wr32(w, o[0], _u32(self._mix(w, o, mods)))

    def op_v_fma_mixlo_f16(self, w, o, mods):
        with np.errstate(over="ignore"):
            h = self._mix(w, o, mods).astype(np.float16).view(np.uint16).astype(np.uint32)
        old = self._v(w, o[0]).astype(np.uint32)
        self.wr32(w, o[0], (old & 0xFFFF0000) | h)

    def op_v_fma_mixhi_f16(self, w, o, mods):
        with np.errstate(over="ignore"):
            h = self._mix(w, o, mods).astype(np.float16).view(np.uint16).astype(np.uint32)
        old = self._v(w, o[0]).astype(np.uint32)
        self.wr32(w, o[0], (old & 0xFFFF) | (h << 16))

    def op_v_mfma_f32_16x16x32_f16(self, w, o, mods):
        def frag(op):     # [lane][8] halves
            base = op[1] + (256 if op[0] == "a" else 0)
            return w.v[base:base + 4].T.copy().view(np.float16).reshape(NLANE, 8).astype(np.float64)
        fa, fb = frag(o[1]), frag(o[2])
        lane = np.arange(NLANE)
        n, q = lane & 15, lane >> 4
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for l in range(NLANE):
            Am[n[l], 8 * q[l]:8 * q[l] + 8] = fa[l]
            Bm[8 * q[l]:8 * q[l] + 8, n[l]] = fb[l]
        D = Am @ Bm
        if o[3][0] in ("a", "v"):
            cb = o[3][1] + (256 if o[3][0] == "a" else 0)
            C = _f32(w.v[cb:cb + 4].reshape(-1)).reshape(4, NLANE)
        else:
            C = np.zeros((4, NLANE), np.float32)
        db = o[0][1] + (256 if o[0][0] == "a" else 0)
        out = np.zeros((4, NLANE), np.float32)
        for i in range(4):
            out[i] = (D[4 * q + i, n] + C[i].astype(np.float64)).astype(np.float32)
        w.v[db:db + 4] = out.view(np.uint32)

    def op_v_mfma_f32_32x32x16_f16(self, w, o, mods):
        def frag(op):     # [lane][8] halves
            base = op[1] + (256 if op[0] == "a" else 0)
            return w.v[base:base + 4].T.copy().view(np.float16).reshape(NLANE, 8).astype(np.float64)
        fa, fb = frag(o[1]), frag(o[2])
        lane = np.arange(NLANE)
        n, h = lane & 31, lane >> 5
        Am = np.zeros((32, 16))
        Bm = np.zeros((16, 32))
        for l in range(NLANE):
            Am[n[l], 8 * h[l]:8 * h[l] + 8] = fa[l]
            Bm[8 * h[l]:8 * h[l] + 8, n[l]] = fb[l]
        D = Am @ Bm
        if o[3][0] in ("a", "v"):
            cb = o[3][1] + (256 if o[3][0] == "a" else 0)
            C = _f32(w.v[cb:cb + 16].reshape(-1)).reshape(16, NLANE)
        else:
            C = np.zeros((16, NLANE), np.float32)
        db = o[0][1] + (256 if o[0][0] == "a" else 0)
        out = np.zeros((16, NLANE), np.float32)
        for i in range(16):
            row = 8 * (i >> 2) + 4 * h + (i & 3)
            out[i] = (D[row, n] + C[i].astype(np.float64)).astype(np.float32)
        w.v[db:db + 16] = out.view(np.uint32)

    def op_v_permlane32_swap_b32(self, w, o, mods):
        x = self._v(w, o[0]).astype(np.uint32).copy()
        y = self._v(w, o[1]).astype(np.uint32).copy()
        nx, ny = x.copy(), y.copy()
        nx[32:64] = y[0:32]
        ny[0:32] = x[32:64]
        full = np.ones(NLANE, bool)
        self.wr32(w, o[0], nx, mask=full)
        self.wr32(w, o[1], ny, mask=full)

    def op_v_cvt_pk_f16_f32(self, w, o, mods):
        with np.errstate(over="ignore"):
            lo = self._fl(w, o[1]).astype(np.float16).view(np.uint16).astype(np.uint32)
            hi = self._fl(w, o[2]).astype(np.float16).view(np.uint16).astype(np.uint32)
        self.wr32(w, o[0], lo | (hi << 16))

    def op_v_mul_f32(self, w, o, mods):
        self.wr32(w, o[0], _u32((self._fl(w, o[1]).astype(np.float64) * self._fl(w, o[2]).astype(np.float64)).astype(np.float32)))

    # ---- LDS / memory ---------------------------------------------------------------------------
    def op_ds_read_b128(self, w, o, mods):
        addr = self._v(w, o[1]).astype(np.int64) + mods.get("offset", 0)
        base = o[0][1] + (256 if o[0][0] == "a" else 0)
        act = self.lanes(w)
        for l in range(NLANE):
            if act[l]:
                a = int(addr[l])
                if a + 16 > len(self.lds):
                    raise RuntimeError(f"LDS read out of bounds {a}")
                w.v[base:base + 4, l] = self.lds[a:a + 16].view(np.uint32)

    def op_ds_write_b128(self, w, o, mods):
        addr = self._v(w, o[0]).astype(np.int64) + mods.get("offset", 0)
        base = o[1][1] + (256 if o[1][0] == "a" else 0)
        act = self.lanes(w)
        for l in range(NLANE):
            if act[l]:
                a = int(addr[l])
                if a + 16 > len(self.lds) or a % 16:
                    raise RuntimeError(f"LDS write out of bounds / misaligned {a}")
                self.lds[a:a + 16] = w.v[base:base + 4, l].copy().view(np.uint8)

    def _gaddr(self, w, o_v, o_s):
        if o_s[0] == "off":
            lo = self._v(w, o_v).astype(np.int64)
            hi = self._v(w, ("v", o_v[1] + 1, 1)).astype(np.int64)
            return lo | (hi << 32)
        return self._v(w, o_v).astype(np.int64) + self.s64(w, o_s)

    def op_global_load_dwordx4(self, w, o, mods):
        addr = self._gaddr(w, o[1], o[2]) + mods.get("offset", 0)
        base = o[0][1] + (256 if o[0][0] == "a" else 0)
        act = self.lanes(w)
        for l in range(NLANE):
            if act[l]:
                w.v[base:base + 4, l] = self.mem.read(int(addr[l]), 16).view(np.uint32)

    def op_global_store_dwordx4(self, w, o, mods, n=4):
        addr = self._gaddr(w, o[0], o[2]) + mods.get("offset", 0)
        base = o[1][1] + (256 if o[1][0] == "a" else 0)
        act = self.lanes(w)
        for l in range(NLANE):
            if act[l]:
                self.mem.write(int(addr[l]), w.v[base:base + n, l].copy().view(np.uint8))

    def op_global_store_dwordx2(self, w, o, mods):
        self.op_global_store_dwordx4(w, o, mods, 2)

    def op_global_store_dword(self, w, o, mods):
        self.op_global_store_dwordx4(w, o, mods, 1)

    def op_global_load_lds_dwordx4(self, w, o, mods):
        addr = self._gaddr(w, o[0], o[1]) + mods.get("offset", 0)
        for l in range(NLANE):
            d = w.m0 + 16 * l
            self.lds[d:d + 16] = self.mem.read(int(addr[l]), 16)
