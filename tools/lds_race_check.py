"""LDS race check of the fused trunk kernels by emulation of their gfx950 ISA (CPU only).

The trunk kernels (chessrl_amd/csrc/tower_x16.hpp) keep two hand-counted pipelines:

  * weight and bias tiles travel L2 -> LDS by LDS-DMA (``global_load_lds_*``); a ring slot may be READ only after
    ``s_waitcnt vmcnt(N)`` has retired the transfers of the issuing wave AND an ``s_barrier`` has published
    them to the other waves, and may be OVERWRITTEN only after a barrier behind its last reader;
  * the MFMA fragments are inline-asm ``ds_read_b128`` that hipcc does not see; their destination registers
    are valid only behind the hand-placed ``s_waitcnt lgkmcnt(N)``.

tools/check_asm_hazards.py walks the control-flow graph for the second kind.  This tool EXECUTES the kernel:
an emulator of the integer / address / control subset of the ISA (floating point, MFMA and loaded data are
"unknown" values, which no address or branch may depend on) runs all 8 waves of one workgroup through the real
instruction stream for a given number of residual blocks and records, per wave and barrier epoch, every LDS
byte read (ds_read), written (ds_write) or in flight by DMA (from the issue of a global_load_lds to the vmcnt
wait that retires it).  Reported:

  1. an LDS byte that is the target of an in-flight DMA transfer in an epoch in which any wave reads or writes
     it, or two transfers in flight to one byte (slot recycled too early / read before it was published);
  2. an LDS byte written by one wave and read or written by another in the same epoch (missing barrier);
  3. an instruction that reads OR OVERWRITES a register while an inline-asm ds_read into it is still in flight
     (exact on the executed path: no path-feasibility guesswork);
  4. an address or branch that depends on an unknown value (the emulation would be meaningless).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --offload-device-only \\
          chessrl_amd/csrc/api.hip -o api.s ;  python tools/lds_race_check.py api.s [n_blocks=2] [kernel filter]
"""
import re
import sys

import numpy as np

U32 = np.uint32
MASK32 = 0xFFFFFFFF
VCC, M0, EXEC = 106, 124, 126
KERNARG = {0x00: 0x10000000, 0x08: 0x20000000, 0x10: 0x30000000, 0x18: 0x40000000,     # planes, wts, bias, out
           0x28: 0x50000000, 0x30: 0x60000000, 0x38: 0x70000000}                         # head_w, head_b, head_out
KERNARG_BASE = 0x7F000000
LIST_BASE = 0x48000000      # IDX kernels: their `out` slot holds the list [n listed, -, boards...]


class EmuError(RuntimeError):
    pass


def _sreg(tok):
    if tok == "vcc":
        return VCC, 2
    if tok == "vcc_lo":
        return VCC, 1
    if tok == "vcc_hi":
        return VCC + 1, 1
    if tok == "exec":
        return EXEC, 2
    if tok == "exec_lo":
        return EXEC, 1
    if tok == "exec_hi":
        return EXEC + 1, 1
    if tok == "m0":
        return M0, 1
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return int(m.group(1)), 1
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1
    return None


def _vreg(tok):
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return (0 if m.group(1) == "v" else 512) + int(m.group(2)), 1
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return (0 if m.group(1) == "v" else 512) + int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
    return None


def _imm(tok):
    try:
        if tok.lower().startswith(("0x", "-0x")):
            return int(tok, 16)
        return int(tok)
    except ValueError:
        return None


def parse_kernel(lines):
    """-> (instructions [(op, operands, modifiers dict, in_asm, text, line no)], label -> index)"""
    ins, labels, in_asm = [], {}, False
    for n, raw in enumerate(lines):
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        l = raw.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":"):
            labels[l[:-1]] = len(ins)
            continue
        if l.startswith("."):
            continue
        op, _, rest = l.partition(" ")
        toks, mods = [], {}
        for t in rest.replace(",", " ").split():
            if ":" in t and not t.startswith(("s[", "v[", "a[")):
                k, _, v = t.partition(":")
                mods[k] = v
            else:
                toks.append(t)
        ins.append((op, toks, mods, in_asm, l, n))
    return ins, labels


class Wave(object):
    def __init__(self, ins, labels, wave, n_blocks, wg=0, want_out=False, want_heads=True, listed=None, kernarg=None):
        self.ins, self.labels, self.wave = ins, labels, wave
        self.s = [None] * 128                     # None = unknown
        self.v = np.zeros((1024, 64), dtype=U32)  # v0..v511, a0..a511
        self.vk = np.zeros((1024, 64), dtype=bool)  # value known, per lane (writes under a partial EXEC mix)
        self.scc = None
        self.s[0], self.s[1] = KERNARG_BASE, 0
        self.s[2] = wg
        self.s[EXEC], self.s[EXEC + 1] = MASK32, MASK32
        self.v[0] = np.arange(64, dtype=U32) + 64 * wave
        self.vk[0] = True
        self._all, self._none = np.ones(64, bool), np.zeros(64, bool)
        self.args = dict(KERNARG)
        if not want_out:
            self.args[0x18] = 0
        self.listed = listed                      # IDX kernels: how many boards the list holds
        if listed is not None:
            self.args[0x18] = LIST_BASE
        if not want_heads:
            self.args[0x38] = 0
        self.int_args = {0x20: n_blocks}          # k_trunk_x16: int n_blocks behind the four pointers
        if kernarg is not None:                   # another kernel family: its own pointer arguments, no integer ones
            self.args, self.int_args = dict(kernarg), {}
        self.n_blocks = n_blocks
        self.events = []                          # (kind, epoch, ...)
        self.epoch = 0
        self.vm = []                              # outstanding vector-memory ops: None or index into events of a DMA
        self.lgkm = []                            # outstanding LGKM ops: (dest registers frozenset or empty, text)
        self.findings = []
        self.n_exec = 0

    # ---- register access -------------------------------------------------------------------------
    def exec_mask(self):
        lo, hi = self.s[EXEC], self.s[EXEC + 1]
        if lo is None or hi is None:
            raise EmuError("EXEC unknown")
        bits = lo | (hi << 32)
        return np.array([(bits >> i) & 1 for i in range(64)], dtype=bool)

    def sget(self, tok, width=1):
        """scalar source -> python int (width 1 or 2 dwords) or None"""
        if tok == "scc":
            return self.scc
        r = _sreg(tok)
        if r is not None:
            idx, n = r
            if width == 2 and n == 2:
                lo, hi = self.s[idx], self.s[idx + 1]
                return None if lo is None or hi is None else lo | (hi << 32)
            if width == 2 and n == 1:
                raise EmuError("64-bit read of a 32-bit register " + tok)
            return self.s[idx]
        im = _imm(tok)
        if im is not None:
            return im & ((1 << (32 * width)) - 1)
        if tok in ("src_scc",):
            return self.scc
        raise EmuError("scalar operand " + tok)

    def sset(self, tok, val, width=1):
        idx, n = _sreg(tok)
        if width == 2:
            self.s[idx] = None if val is None else val & MASK32
            self.s[idx + 1] = None if val is None else (val >> 32) & MASK32
        else:
            self.s[idx] = None if val is None else val & MASK32

    def vsrc(self, tok, comp=0):
        """vector-ALU source as (uint32[64] array, known): VGPR dword `comp`, SGPR or constant"""
        r = _vreg(tok)
        if r is not None:
            self.touch_read(r[0], r[1])
            return self.v[r[0] + comp], self.vk[r[0] + comp]
        rs = _sreg(tok)
        if rs is not None:
            val = self.s[rs[0] + comp] if comp < rs[1] else 0
            if val is None:
                return np.zeros(64, U32), self._none
            return np.full(64, val & MASK32, U32), self._all
        im = _imm(tok)
        if im is not None:
            full = im & 0xFFFFFFFFFFFFFFFF if im < 0 else im
            return np.full(64, (full >> (32 * comp)) & MASK32, U32), self._all
        if re.fullmatch(r"-?\d+\.\d*", tok) or tok in ("src_shared_base",):
            return np.zeros(64, U32), self._none
        raise EmuError("vector operand " + tok)

    def vsrc64(self, tok):
        lo, k0 = self.vsrc(tok, 0)
        r, rs = _vreg(tok), _sreg(tok)
        if r is not None and r[1] < 2 or rs is not None and rs[1] < 2:
            raise EmuError("64-bit read of a 32-bit operand " + tok)
        hi, k1 = self.vsrc(tok, 1)
        return lo.astype(np.uint64) | (hi.astype(np.uint64) << np.uint64(32)), k0 & k1

    def vdst(self, tok, vals, known, comp=0):
        r = _vreg(tok)
        self.touch_write(r[0] + comp, 1)
        m = self.exec_mask()
        known = self._kmask(known)
        self.v[r[0] + comp][m] = vals[m] if isinstance(vals, np.ndarray) else vals
        self.vk[r[0] + comp][m] = known[m]

    def _kmask(self, k):
        if isinstance(k, np.ndarray):
            return k
        return self._all if k else self._none

    def _known(self, k):
        """every ACTIVE lane's value is known"""
        return bool(self._kmask(k)[self.exec_mask()].all())

    def vdst64(self, tok, vals, known):
        self.vdst(tok, (vals & np.uint64(MASK32)).astype(U32), known, 0)
        self.vdst(tok, (vals >> np.uint64(32)).astype(U32), known, 1)

    def vunknown(self, tok):
        r = _vreg(tok)
        if r is None:
            rs = _sreg(tok)
            if rs is not None:
                for i in range(rs[1]):
                    self.s[rs[0] + i] = None
            return
        self.touch_write(r[0], r[1])
        m = self.exec_mask()
        for i in range(r[1]):
            self.vk[r[0] + i][m] = False

    # ---- in-flight inline-asm ds_read destinations ---------------------------------------------------
    def touch_read(self, idx, n):
        for dest, text in self.lgkm:
            if dest and any(idx <= d < idx + n for d in dest):
                self.findings.append(("reg", self.cur_line, "READS a register with an asm ds_read in flight: " + self.cur_text))
                return

    def touch_write(self, idx, n):
        for dest, text in self.lgkm:
            if dest and any(idx <= d < idx + n for d in dest):
                self.findings.append(("reg", self.cur_line, "OVERWRITES a register with an asm ds_read in flight: " + self.cur_text))
                return

    # ---- LDS events ----------------------------------------------------------------------------------
    def lds_bytes(self, addr, size, lanes=None):
        m = self.exec_mask() if lanes is None else lanes
        a = addr[m].astype(np.int64)
        return (a[:, None] + np.arange(size, dtype=np.int64)[None, :]).ravel()

    # ---- execution -----------------------------------------------------------------------------------
    def run(self, max_instructions=3000000):
        pc = 0
        ins = self.ins
        while True:
            if self.n_exec > max_instructions:
                raise EmuError("instruction budget exceeded")
            self.n_exec += 1
            op, t, mods, in_asm, text, line = ins[pc]
            self.cur_text, self.cur_line = text, line
            nxt = pc + 1
            if op == "s_endpgm":
                break
            elif op.startswith("s_cbranch") or op == "s_branch":
                if op == "s_branch":
                    cond = True
                elif op in ("s_cbranch_scc0", "s_cbranch_scc1"):
                    if self.scc is None:
                        raise EmuError("branch on unknown SCC: " + text)
                    cond = (self.scc == 1) == (op == "s_cbranch_scc1")
                elif op in ("s_cbranch_vccz", "s_cbranch_vccnz"):
                    vcc = self.sget("vcc", 2)
                    if vcc is None:
                        raise EmuError("branch on unknown VCC: %s (line %d)" % (text, line))
                    cond = (vcc == 0) == (op == "s_cbranch_vccz")
                elif op in ("s_cbranch_execz", "s_cbranch_execnz"):
                    ex = self.sget("exec", 2)
                    cond = (ex == 0) == (op == "s_cbranch_execz")
                else:
                    raise EmuError("branch " + op)
                if cond:
                    nxt = self.labels[t[0]]
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m:
                    keep = int(m.group(1))
                    while len(self.vm) > keep:
                        e = self.vm.pop(0)
                        if e is not None:
                            self.events[e][3] = self.epoch             # retired in this epoch
                m = re.search(r"lgkmcnt\((\d+)\)", text)
                if m:
                    keep = int(m.group(1))
                    if keep < len(self.lgkm):
                        self.lgkm = self.lgkm[len(self.lgkm) - keep:] if keep else []
            elif op == "s_barrier":
                self.events.append(["barrier", self.epoch])
                self.epoch += 1
            elif op in ("s_nop", "s_setprio", "s_sleep", "s_sethalt", "s_waitcnt_depctr", "s_setreg_imm32_b32"):
                pass
            elif op.startswith("s_"):
                self.salu(op, t, mods, text)
            elif op.startswith("ds_"):
                self.ds(op, t, mods, in_asm, text)
            elif op.startswith("scratch_"):
                self.scratch(op, t, mods, text)
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
                self.vmem(op, t, mods, text)
            elif op.startswith("v_"):
                self.valu(op, t, mods, text)
            else:
                raise EmuError("opcode " + op)
            pc = nxt
        return self

    # ---- scalar ALU ----------------------------------------------------------------------------------
    def salu(self, op, t, mods, text):
        def s32(x):
            return x - (1 << 32) if x is not None and x & 0x80000000 else x
        g = self.sget
        if op in ("s_mov_b32", "s_mov_b64"):
            w = 2 if op.endswith("64") else 1
            self.sset(t[0], g(t[1], w), w)
        elif op == "s_movk_i32":
            self.sset(t[0], _imm(t[1]) & MASK32 if _imm(t[1]) >= 0 else (_imm(t[1]) + (1 << 32)))
            if _imm(t[1]) & 0x8000 and _imm(t[1]) > 0:
                self.sset(t[0], (_imm(t[1]) | 0xFFFF0000) & MASK32)
        elif op in ("s_add_i32", "s_add_u32", "s_sub_i32", "s_sub_u32", "s_addc_u32", "s_addk_i32"):
            a, b = (g(t[0]), _imm(t[1]) & MASK32) if op == "s_addk_i32" else (g(t[1]), g(t[2]))
            if a is None or b is None or (op == "s_addc_u32" and self.scc is None):
                self.sset(t[0], None)
                self.scc = None
            else:
                if op.startswith("s_sub"):
                    full = a - b
                    self.scc = int(full < 0) if op == "s_sub_u32" else int(not -(1 << 31) <= s32(a) - s32(b) < (1 << 31))
                else:
                    full = a + b + (self.scc if op == "s_addc_u32" else 0)
                    self.scc = (int(full > MASK32) if op in ("s_add_u32", "s_addc_u32")
                                else int(not -(1 << 31) <= s32(a) + s32(b) < (1 << 31)))
                self.sset(t[0], full & MASK32)
        elif op in ("s_bitset1_b32", "s_bitset0_b32"):
            a, b = g(t[0]), g(t[1])
            if a is None or b is None:
                self.sset(t[0], None)
            else:
                self.sset(t[0], (a | (1 << (b & 31))) if op == "s_bitset1_b32" else (a & ~(1 << (b & 31)) & MASK32))
        elif op == "s_mulk_i32":                             # D = D * signext(simm16); SCC unchanged
            a, k = g(t[0]), _imm(t[1]) & 0xFFFF
            k = k - 0x10000 if k & 0x8000 else k
            self.sset(t[0], None if a is None else (s32(a) * k) & MASK32)
        elif op in ("s_mul_i32", "s_mul_hi_u32", "s_mul_hi_i32"):
            a, b = g(t[1]), g(t[2])
            if a is None or b is None:
                self.sset(t[0], None)
            elif op == "s_mul_i32":
                self.sset(t[0], (a * b) & MASK32)
            elif op == "s_mul_hi_u32":
                self.sset(t[0], ((a * b) >> 32) & MASK32)
            else:
                self.sset(t[0], ((s32(a) * s32(b)) >> 32) & MASK32)
        elif op in ("s_lshl_b32", "s_lshr_b32", "s_ashr_i32", "s_lshl_b64", "s_lshr_b64"):
            w = 2 if op.endswith("64") else 1
            a, b = g(t[1], w), g(t[2])
            if a is None or b is None:
                self.sset(t[0], None, w)
                self.scc = None
            else:
                sh = b & (63 if w == 2 else 31)
                if op.startswith("s_lshl"):
                    r = (a << sh) & ((1 << (32 * w)) - 1)
                elif op.startswith("s_lshr"):
                    r = a >> sh
                else:
                    r = (s32(a) >> sh) & MASK32
                self.sset(t[0], r, w)
                self.scc = int(r != 0)
        elif op in ("s_and_b32", "s_or_b32", "s_xor_b32", "s_andn2_b32", "s_orn2_b32",
                    "s_and_b64", "s_or_b64", "s_xor_b64", "s_andn2_b64", "s_orn2_b64"):
            w = 2 if op.endswith("64") else 1
            a, b = g(t[1], w), g(t[2], w)
            full = (1 << (32 * w)) - 1
            if a is None or b is None:
                # x & 0 and x | ~0 are known whatever x is
                if op.startswith("s_and_b") and (a == 0 or b == 0):
                    r = 0
                elif op.startswith("s_andn2") and (a == 0 or b == full):
                    r = 0
                else:
                    r = None
            else:
                kind = op[2:].split("_b")[0]
                r = {"and": a & b, "or": a | b, "xor": a ^ b, "andn2": a & ~b & full, "orn2": (a | ~b) & full}[kind]
            self.sset(t[0], r, w)
            self.scc = None if r is None else int(r != 0)
        elif op in ("s_not_b32", "s_not_b64"):
            w = 2 if op.endswith("64") else 1
            a = g(t[1], w)
            r = None if a is None else ~a & ((1 << (32 * w)) - 1)
            self.sset(t[0], r, w)
            self.scc = None if r is None else int(r != 0)
        elif op in ("s_max_i32", "s_min_i32", "s_max_u32", "s_min_u32"):
            a, b = g(t[1]), g(t[2])
            if a is None or b is None:
                self.sset(t[0], None)
                self.scc = None
            else:
                x, y = (s32(a), s32(b)) if op.endswith("i32") else (a, b)
                first = x > y if "max" in op else x < y
                self.sset(t[0], a if first else b)
                self.scc = int(first)
        elif op in ("s_cselect_b32", "s_cselect_b64"):
            w = 2 if op.endswith("64") else 1
            if self.scc is None:
                self.sset(t[0], None, w)
            else:
                self.sset(t[0], g(t[1], w) if self.scc else g(t[2], w), w)
        elif op.startswith("s_cmp_") or op.startswith("s_cmpk_"):
            m = re.fullmatch(r"s_cmpk?_(eq|lg|gt|ge|lt|le)_(i32|u32|u64)", op)
            if not m:
                raise EmuError("compare " + op)
            w = 2 if m.group(2) == "u64" else 1
            a, b = g(t[0], w), g(t[1], w)
            if a is None or b is None:
                self.scc = None
            else:
                if m.group(2) == "i32":
                    a, b = s32(a & MASK32), s32(b & MASK32)
                self.scc = int({"eq": a == b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[m.group(1)])
        elif op in ("s_bitcmp1_b32", "s_bitcmp0_b32"):
            a, b = g(t[0]), g(t[1])
            self.scc = None if a is None or b is None else int(((a >> (b & 31)) & 1) == (1 if op == "s_bitcmp1_b32" else 0))
        elif op == "s_and_saveexec_b64":
            ex, a = g("exec", 2), g(t[1], 2)
            if a is None:
                raise EmuError("EXEC from an unknown mask: " + text)
            self.sset(t[0], ex, 2)
            self.sset("exec", ex & a, 2)
            self.scc = int((ex & a) != 0)
        elif op == "s_or_saveexec_b64":
            ex, a = g("exec", 2), g(t[1], 2)
            if a is None:
                raise EmuError("EXEC from an unknown mask: " + text)
            self.sset(t[0], ex, 2)
            self.sset("exec", ex | a, 2)
            self.scc = int((ex | a) != 0)
        elif op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[op]
            base, off = g(t[1], 2), _imm(t[2]) if len(t) > 2 else 0
            idx, _ = _sreg(t[0])
            for i in range(n):
                val = None
                if self.listed is not None and base is not None and LIST_BASE <= base + off + 4 * i < LIST_BASE + 4096:
                    word = (base + off + 4 * i - LIST_BASE) // 4
                    # [0] listed boards, [1..3] statistics words, [LIST_HEADER = 4 ...] the boards
                    val = self.listed if word == 0 else (0 if word < 4 else (7 * (word - 4) + 3) & MASK32)
                elif base == KERNARG_BASE:
                    o = off + 4 * i
                    if o in self.int_args:
                        val = self.int_args[o]
                    elif (o & ~7) in self.args:
                        val = (self.args[o & ~7] >> (32 * ((o >> 2) & 1))) & MASK32
                self.s[idx + i] = val
            self.lgkm.append((frozenset(), text))
        else:
            raise EmuError("scalar opcode " + op)

    # ---- LDS -----------------------------------------------------------------------------------------
    def ds(self, op, t, mods, in_asm, text):
        off = int(mods.get("offset", "0"), 0)
        if op in ("ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_read2_b32", "ds_read2_b64", "ds_read_u16", "ds_read_u8"):
            addr, known = self.vsrc(t[1])
            if not self._known(known):
                raise EmuError("ds_read from an unknown address: " + text)
            d = _vreg(t[0])
            if op.startswith("ds_read2"):
                sz = 4 if op.endswith("b32") else 8
                for o in (int(mods.get("offset0", "0"), 0), int(mods.get("offset1", "0"), 0)):
                    self.events.append(["read", self.epoch, self.lds_bytes(addr + U32(o * sz), sz), text])
            else:
                sz = {"ds_read_b128": 16, "ds_read_b64": 8, "ds_read_b32": 4, "ds_read_u16": 2, "ds_read_u8": 1}[op]
                self.events.append(["read", self.epoch, self.lds_bytes(addr + U32(off), sz), text])
            self.vunknown(t[0])
            self.lgkm.append((frozenset(range(d[0], d[0] + d[1])) if in_asm else frozenset(), text))
        elif op in ("ds_write_b128", "ds_write_b64", "ds_write_b32", "ds_write_b16", "ds_write_b8", "ds_write2_b32", "ds_write2_b64"):
            addr, known = self.vsrc(t[0])
            if not self._known(known):
                raise EmuError("ds_write to an unknown address: " + text)
            for tok in t[1:]:
                self.vsrc(tok)                                           # data registers are read
            if op.startswith("ds_write2"):
                sz = 4 if op.endswith("b32") else 8
                for o in (int(mods.get("offset0", "0"), 0), int(mods.get("offset1", "0"), 0)):
                    self.events.append(["write", self.epoch, self.lds_bytes(addr + U32(o * sz), sz), text])
            else:
                sz = {"ds_write_b128": 16, "ds_write_b64": 8, "ds_write_b32": 4, "ds_write_b16": 2, "ds_write_b8": 1}[op]
                self.events.append(["write", self.epoch, self.lds_bytes(addr + U32(off), sz), text])
            self.lgkm.append((frozenset(), text))
        elif op in ("ds_bpermute_b32", "ds_permute_b32", "ds_swizzle_b32"):
            self.vsrc(t[1])
            self.vunknown(t[0])
            self.lgkm.append((frozenset(), text))
        else:
            raise EmuError("LDS opcode " + op)

    # ---- register spills: private memory per lane, by byte offset ("scratch_store_dwordxN off, v[..], off offset:K") ----
    def scratch(self, op, t, mods, text):
        n = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(op.split("_")[-1])
        if n is None or "off" not in t:
            raise EmuError("scratch opcode " + text)
        mem = self.__dict__.setdefault("_scratch", {})
        off = int(mods.get("offset", "0"), 0)
        if op.startswith("scratch_store"):
            r = _vreg([x for x in t if _vreg(x) is not None][0])
            self.touch_read(r[0], r[1])
            for i in range(n):
                mem[off + 4 * i] = (self.v[r[0] + i].copy(), self.vk[r[0] + i].copy())
        else:
            for i in range(n):
                vals, known = mem.get(off + 4 * i, (np.zeros(64, U32), self._none))
                self.vdst(t[0], vals, known, comp=i)
        self.vm.append(None)

    # ---- vector memory --------------------------------------------------------------------------------
    def vmem(self, op, t, mods, text):
        if op.startswith("global_load_lds_"):
            sz = {"global_load_lds_dword": 4, "global_load_lds_dwordx4": 16, "global_load_lds_dwordx3": 12,
                  "global_load_lds_ushort": 2, "global_load_lds_ubyte": 1}[op]
            m0 = self.s[M0]
            if m0 is None:
                raise EmuError("LDS-DMA with unknown M0: " + text)
            self.vsrc(t[0])
            base = m0 + int(mods.get("offset", "0"), 0)
            lanes = self.exec_mask()
            addr = (base + sz * np.arange(64, dtype=np.int64))[lanes]
            by = (addr[:, None] + np.arange(sz, dtype=np.int64)[None, :]).ravel()
            self.vm.append(len(self.events))
            self.events.append(["dma", self.epoch, by, None, text])      # [3] = epoch of the retiring vmcnt wait
        elif "load" in op:
            self.vsrc(t[1])
            self.vunknown(t[0])
            self.vm.append(None)
        elif "store" in op or "atomic" in op:
            for tok in t[:2]:
                if _vreg(tok) is not None:
                    self.vsrc(tok)
            self.vm.append(None)
        else:
            raise EmuError("memory opcode " + op)

    # ---- vector ALU -----------------------------------------------------------------------------------
    def valu(self, op, t, mods, text):
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        sd = op.endswith("_sdwa")

        def sel(x, which):
            s = mods.get(which, "DWORD")
            if s == "DWORD":
                return x
            if s.startswith("BYTE_"):
                return (x >> U32(8 * int(s[-1]))) & U32(0xFF)
            if s.startswith("WORD_"):
                return (x >> U32(16 * int(s[-1]))) & U32(0xFFFF)
            raise EmuError("sdwa select " + s)

        def src(i, which=None):
            x, k = self.vsrc(t[i])
            if sd and which:
                x = sel(x, which)
            return x, k

        u64 = np.uint64
        if base == "v_mov_b32":
            x, k = src(1)
            self.vdst(t[0], x, k)
        elif base == "v_mov_b64":
            x, k = self.vsrc64(t[1]) if (_vreg(t[1]) or _sreg(t[1])) else (np.full(64, _imm(t[1]) & 0xFFFFFFFFFFFFFFFF, u64), self._all)
            self.vdst64(t[0], x, k)
        elif base in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mul_lo_u32",
                      "v_mul_u32_u24", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32",
                      "v_mul_lo_u16", "v_sub_u16", "v_add_u16", "v_lshrrev_b16", "v_lshlrev_b16", "v_mul_hi_u32"):
            a, ka = src(1, "src0_sel")
            b, kb = src(2, "src1_sel")
            k = ka & kb
            if base == "v_and_b32":
                r = a & b
                k = k | (ka & (a == 0)) | (kb & (b == 0))
            elif base == "v_or_b32":
                r = a | b
            elif base == "v_xor_b32":
                r = a ^ b
            elif base == "v_add_u32":
                r = a + b
            elif base == "v_sub_u32":
                r = a - b
            elif base == "v_subrev_u32":
                r = b - a
            elif base == "v_mul_lo_u32":
                r = (a.astype(u64) * b.astype(u64)).astype(U32)
            elif base == "v_mul_hi_u32":
                r = ((a.astype(u64) * b.astype(u64)) >> u64(32)).astype(U32)
            elif base == "v_mul_u32_u24":
                r = ((a & U32(0xFFFFFF)).astype(u64) * (b & U32(0xFFFFFF)).astype(u64)).astype(U32)
            elif base == "v_lshlrev_b32":
                r = b << (a & U32(31))
            elif base == "v_lshrrev_b32":
                r = b >> (a & U32(31))
            elif base == "v_ashrrev_i32":
                r = (b.astype(np.int32) >> (a & U32(31)).astype(np.int32)).astype(U32)
            elif base == "v_min_u32":
                r = np.minimum(a, b)
            elif base == "v_max_u32":
                r = np.maximum(a, b)
            elif base == "v_min_i32":
                r = np.minimum(a.astype(np.int32), b.astype(np.int32)).astype(U32)
            elif base == "v_max_i32":
                r = np.maximum(a.astype(np.int32), b.astype(np.int32)).astype(U32)
            elif base == "v_mul_lo_u16":
                r = (a * b) & U32(0xFFFF)
            elif base == "v_sub_u16":
                r = (a - b) & U32(0xFFFF)
            elif base == "v_add_u16":
                r = (a + b) & U32(0xFFFF)
            elif base == "v_lshrrev_b16":
                r = (b & U32(0xFFFF)) >> (a & U32(15))
            else:
                r = (b << (a & U32(15))) & U32(0xFFFF)
            if sd and mods.get("dst_sel", "DWORD") != "DWORD":
                raise EmuError("sdwa dst_sel: " + text)
            self.vdst(t[0], r, k)
        elif base in ("v_lshl_add_u32", "v_lshl_or_b32", "v_add3_u32", "v_or3_b32", "v_and_or_b32", "v_mad_u32_u24",
                      "v_bfe_u32", "v_add_lshl_u32", "v_xad_u32", "v_bitop3_b32", "v_mad_u32_u16", "v_alignbit_b32", "v_perm_b32"):
            a, ka = src(1)
            b, kb = src(2)
            c, kc = src(3)
            k = ka & kb & kc
            if base == "v_lshl_add_u32":
                r = (a << (b & U32(31))) + c
            elif base == "v_lshl_or_b32":
                r = (a << (b & U32(31))) | c
            elif base == "v_add3_u32":
                r = a + b + c
            elif base == "v_or3_b32":
                r = a | b | c
            elif base == "v_and_or_b32":
                r = (a & b) | c
            elif base == "v_mad_u32_u24":
                r = ((a & U32(0xFFFFFF)).astype(u64) * (b & U32(0xFFFFFF)).astype(u64)).astype(U32) + c
            elif base == "v_mad_u32_u16":
                r = (a & U32(0xFFFF)) * (b & U32(0xFFFF)) + c
            elif base == "v_bfe_u32":
                r = (a >> (b & U32(31))) & ((U32(1) << (c & U32(31))) - U32(1))
            elif base == "v_add_lshl_u32":
                r = (a + b) << (c & U32(31))
            elif base == "v_xad_u32":
                r = (a ^ b) + c
            elif base == "v_bitop3_b32":
                tt = int(mods["bitop3"], 0)
                r = np.zeros(64, U32)
                for idx in range(8):
                    if (tt >> idx) & 1:
                        r |= ((a if idx & 4 else ~a) & (b if idx & 2 else ~b) & (c if idx & 1 else ~c))
            else:
                r, k = a, self._none
            self.vdst(t[0], r, k)
        elif base == "v_lshl_add_u64":
            a, ka = self.vsrc64(t[1]) if _imm(t[1]) is None else (np.full(64, _imm(t[1]), u64), self._all)
            b, kb = src(2)
            c, kc = self.vsrc64(t[3]) if _imm(t[3]) is None else (np.full(64, _imm(t[3]) & 0xFFFFFFFFFFFFFFFF, u64), self._all)
            self.vdst64(t[0], (a << (b.astype(u64) & u64(7))) + c, ka & kb & kc)
        elif base in ("v_lshlrev_b64", "v_lshrrev_b64"):
            a, ka = src(1)
            b, kb = self.vsrc64(t[2])
            sh = a.astype(u64) & u64(63)
            self.vdst64(t[0], (b << sh) if base == "v_lshlrev_b64" else (b >> sh), ka & kb)
        elif base in ("v_mad_u64_u32", "v_mad_i64_i32"):
            a, ka = src(2)
            b, kb = src(3)
            if base == "v_mad_i64_i32":                              # signed 32 x 32 -> 64
                a = a.astype(np.int32).astype(np.int64).astype(u64)
                b = b.astype(np.int32).astype(np.int64).astype(u64)
            c, kc = (np.full(64, _imm(t[4]) & 0xFFFFFFFFFFFFFFFF, u64), self._all) if _imm(t[4]) is not None else self.vsrc64(t[4])
            self.vdst64(t[0], a.astype(u64) * b.astype(u64) + c, ka & kb & kc)
            self.vunknown(t[1])
        elif base in ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32"):
            a, ka = src(2)
            b, kb = src(3)
            k = ka & kb
            cin = np.zeros(64, u64)
            if base in ("v_addc_co_u32", "v_subb_co_u32"):
                cm = self.sget(t[4], 2)
                k = k & self._kmask(cm is not None)
                cin = np.array([((cm or 0) >> i) & 1 for i in range(64)], dtype=u64)
            if base.startswith("v_add"):
                full = a.astype(u64) + b.astype(u64) + cin
                carry = full >> u64(32)
            else:
                x, y = (b, a) if base == "v_subrev_co_u32" else (a, b)
                full = (x.astype(np.int64) - y.astype(np.int64) - cin.astype(np.int64))
                carry = (full < 0).astype(u64)
                full = full.astype(u64)
            self.vdst(t[0], (full & u64(MASK32)).astype(U32), k)
            m = self.exec_mask()
            old = self.sget(t[1], 2) or 0
            bits = sum(int(carry[i]) << i for i in range(64) if m[i]) | (old & ~sum(1 << i for i in range(64) if m[i]))
            self.sset(t[1], bits if self._known(k) else None, 2)
        elif base.startswith("v_cmp_") or base.startswith("v_cmpx_"):
            m = re.fullmatch(r"v_cmpx?_(eq|ne|lg|gt|ge|lt|le)_(u32|i32|u64|i64|u16|i16)", base)
            if not m:
                self.sset(t[0], None, 2)                                  # float compares: unknown
                return
            if m.group(2) in ("u64", "i64"):
                a, ka = self.vsrc64(t[1]) if _imm(t[1]) is None else (np.full(64, _imm(t[1]) & 0xFFFFFFFFFFFFFFFF, u64), self._all)
                b, kb = self.vsrc64(t[2]) if _imm(t[2]) is None else (np.full(64, _imm(t[2]) & 0xFFFFFFFFFFFFFFFF, u64), self._all)
                if m.group(2) == "i64":
                    a, b = a.astype(np.int64), b.astype(np.int64)
            else:
                a, ka = src(1, "src0_sel")
                b, kb = src(2, "src1_sel")
                if m.group(2) in ("u16", "i16"):
                    a, b = a & U32(0xFFFF), b & U32(0xFFFF)
                if m.group(2) == "i32":
                    a, b = a.astype(np.int32), b.astype(np.int32)
                if m.group(2) == "i16":
                    a, b = a.astype(np.uint16).astype(np.int16), b.astype(np.uint16).astype(np.int16)
            res = {"eq": a == b, "ne": a != b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[m.group(1)]
            em = self.exec_mask()
            bits = sum(1 << i for i in range(64) if res[i] and em[i])
            self.sset(t[0], bits if self._known(ka & kb) else None, 2)
            if base.startswith("v_cmpx"):
                self.sset("exec", bits, 2)
        elif base == "v_cndmask_b32":
            a, ka = src(1)
            b, kb = src(2)
            cm = self.sget(t[3], 2)
            if cm is None:
                self.vdst(t[0], a, self._none)
            else:
                pick = np.array([(cm >> i) & 1 for i in range(64)], dtype=bool)
                self.vdst(t[0], np.where(pick, b, a), np.where(pick, kb, ka))     # a lane is as known as the source it picks
        elif base == "v_readfirstlane_b32":
            x, k = src(1)
            m = self.exec_mask()
            lane = int(np.argmax(m)) if m.any() else 0
            self.sset(t[0], int(x[lane]) if k[lane] else None)
        elif base == "v_readlane_b32":
            x, k = src(1)
            ln = self.sget(t[2])
            self.sset(t[0], int(x[ln & 63]) if ln is not None and k[ln & 63] else None)
        elif base == "v_writelane_b32":
            # vD[lane] = sS (SGPR spills): ignores EXEC
            r = _vreg(t[0])
            val, ln = self.sget(t[1]), self.sget(t[2])
            if ln is None:
                raise EmuError("v_writelane with an unknown lane: " + text)
            self.touch_write(r[0], 1)
            self.v[r[0]][ln & 63] = (val or 0) & MASK32
            self.vk[r[0]][ln & 63] = val is not None
        elif base == "v_bfrev_b32":
            x, k = src(1)
            r = np.array([int("{:032b}".format(int(v))[::-1], 2) for v in x], dtype=U32)
            self.vdst(t[0], r, k)
        elif base in ("v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
            x, k = src(1)
            self.vdst(t[0], x, k)
        elif base == "v_not_b32":
            x, k = src(1)
            self.vdst(t[0], ~x, k)
        else:
            # floating point, conversions, packed math, MFMA: values no address or branch depends on
            if not re.match(r"v_(mfma|add_f|sub_f|mul_f|max_f|min_f|fma|mac_f|cvt|pk_|exp|log|rcp|rsq|sqrt|med3|fmac|dot|"
                            r"max3|min3|mad_f|ldexp|frexp|trunc|floor|ceil|rndne|fract|cndmask|swap|permlane|smfmac)", base):
                raise EmuError("vector opcode %s (line %d)" % (op, self.cur_line))
            for tok in t[1:]:
                if _vreg(tok) is not None:
                    self.vsrc(tok)
            self.vunknown(t[0])


def check_workgroup(ins, labels, n_blocks, lds_bytes=160 * 1024, want_out=False, listed=None, kernarg=None):
    """Emulate the 8 waves of workgroup 0 and cross-check their LDS traffic epoch by epoch."""
    waves = [Wave(ins, labels, w, n_blocks, want_out=want_out, listed=listed, kernarg=kernarg).run() for w in range(8)]
    findings = []
    for w in waves:
        for kind, line, msg in w.findings:
            findings.append("wave %d line %d: %s" % (w.wave, line, msg))
    nb = [sum(1 for e in w.events if e[0] == "barrier") for w in waves]
    if len(set(nb)) != 1:
        findings.append("waves disagree on the number of barriers: %s" % nb)
        return findings, {"instructions": sum(w.n_exec for w in waves)}
    n_epochs = nb[0] + 1
    per_epoch = [[[] for _ in range(8)] for _ in range(n_epochs)]
    dma_total = 0
    for w in waves:
        if w.vm and any(e is not None for e in w.vm):
            findings.append("wave %d ends with LDS-DMA transfers never waited for" % w.wave)
        for e in w.events:
            if e[0] in ("read", "write"):
                per_epoch[e[1]][w.wave].append(e)
            elif e[0] == "dma":
                dma_total += 1
                last = e[3] if e[3] is not None else n_epochs - 1
                for ep in range(e[1], last + 1):
                    per_epoch[ep][w.wave].append(e)
    size = lds_bytes + 64
    rd, wr, dm = np.zeros(size, np.uint8), np.zeros(size, np.uint8), np.zeros(size, np.uint16)
    seen = set()
    for ep in range(n_epochs):
        rd[:] = 0
        wr[:] = 0
        dm[:] = 0
        who = {}
        for wv in range(8):
            bit = np.uint8(1 << wv)
            for e in per_epoch[ep][wv]:
                by = e[2]
                if by.size and (by.min() < 0 or by.max() >= size):
                    findings.append("wave %d: LDS access outside the allocation: %s" % (wv, e[-1]))
                    continue
                if e[0] == "read":
                    rd[by] |= bit
                elif e[0] == "write":
                    wr[by] |= bit
                else:
                    np.add.at(dm, by, 1)
                who.setdefault(e[0], set()).add(e[-1])
        multi_w = (wr & (wr - np.uint8(1))) != 0
        cross = (wr != 0) & ((rd & ~wr) != 0)
        dma_rw = (dm != 0) & ((rd != 0) | (wr != 0))
        dma_dma = dm > 1
        for name, mask in (("two waves write one LDS byte in one barrier epoch", multi_w),
                           ("a wave reads an LDS byte another wave writes in the same barrier epoch", cross),
                           ("an LDS byte is read or written while an LDS-DMA transfer into it is in flight", dma_rw),
                           ("two LDS-DMA transfers into one LDS byte are in flight together", dma_dma)):
            if mask.any():
                lo = int(np.argmax(mask))
                key = (name, lo // 1024)
                if key not in seen:
                    seen.add(key)
                    findings.append("epoch %d: %s (%d bytes, first at LDS offset 0x%x)" % (ep, name, int(mask.sum()), lo))
    stats = {"instructions": sum(w.n_exec for w in waves), "epochs": n_epochs, "dma_transfers": dma_total,
             "lds_reads": sum(1 for w in waves for e in w.events if e[0] == "read"),
             "lds_writes": sum(1 for w in waves for e in w.events if e[0] == "write")}
    return findings, stats


def kernels_of(path):
    """[(family, template arguments, instruction lines)] of every kernel with hand-counted pipelines: the fused trunk
    k_trunk_x16<...> (tower_x16.hpp) and the layer-wise convolution k_layer_conv<CHUNKS, KIND, IDX> (tower_layer.hpp)."""
    L = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"_ZN9crl_tower\d+k_(trunk_x16|layer_conv)I", l)]
    out = []
    for i, name in starts:
        end = next(j for j in range(i, len(L)) if "s_endpgm" in L[j])
        m = re.search(r"\d+(k_trunk_x16|k_layer_conv)I(.*?)EEv", name)
        tmpl = m.group(2).replace("Li", "").replace("E", ",").rstrip(",")
        out.append((m.group(1), tmpl, L[i + 1:end + 1]))
    return out


# k_layer_conv(act_in, wts, bias, act_out, list, head_w, head_b, head_out, out): pointers only
def layer_kernarg(kind, indexed):
    return {0x00: 0x10000000, 0x08: 0x20000000, 0x10: 0x30000000, 0x18: 0x40000000, 0x20: LIST_BASE if indexed else 0,
            0x28: 0x50000000, 0x30: 0x60000000, 0x38: 0x70000000 if kind >= 3 else 0, 0x40: 0x78000000 if kind == 4 else 0}


def layer_listed(idx):
    """How many boards the emulated list of a k_layer_conv<.., .., IDX, ..> launch holds (None: not indexed)."""
    return {"0": None, "1": 1, "2": 1, "3": 301}[str(idx)]


def check_kernel(args):
    family, tmpl, seg, n_blocks = args
    ins, labels = parse_kernel(seg)
    targs = tmpl.split(",")
    try:
        if family == "k_layer_conv":
            # one launch = one convolution (the whole kernel is executed: 4 or 8 K-chunks x 9 taps); an indexed launch
            # with one listed board runs with its padding (IDX 3 only works on a list beyond IDX_SMALL_MAX = 256 boards)
            indexed = targs[2] != "0"
            findings, stats = check_workgroup(ins, labels, 0, listed=layer_listed(targs[2]),
                                              kernarg=layer_kernarg(int(targs[1]), indexed))
        else:
            # an indexed kernel (8th template argument) takes its boards from a list: one listed board, so that a
            # workgroup of several boards runs with its padding (the last entry repeated)
            indexed = len(targs) > 7 and targs[7] == "1"
            findings, stats = check_workgroup(ins, labels, n_blocks, listed=1 if indexed else None)
    except EmuError as e:
        findings, stats = ["emulation stopped: %s" % e], {}
    return family, tmpl, findings, stats


def main(path, n_blocks=2, only=None, jobs=None):
    import multiprocessing
    import os
    work = [(family, tmpl, seg, n_blocks) for family, tmpl, seg in kernels_of(path)
            if only is None or only in tmpl or only in family]
    jobs = jobs or min(len(work), os.cpu_count() or 1)
    if jobs > 1:
        with multiprocessing.Pool(jobs) as pool:
            results = pool.map(check_kernel, work)
    else:
        results = [check_kernel(w) for w in work]
    total = 0
    for family, tmpl, findings, stats in results:
        what = "%d residual blocks" % n_blocks if family == "k_trunk_x16" else "one convolution"
        print("%s<%s> (%s): %s  %s" % (family, tmpl, what, "ok" if not findings else "%d findings" % len(findings), stats))
        for f in findings[:12]:
            print("      " + f)
        total += len(findings)
    return total


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2,
                       sys.argv[3] if len(sys.argv) > 3 else None) else 0)
