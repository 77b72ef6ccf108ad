"""gcn_emu.py — a small functional emulator for the subset of gfx950 assembly the generated conv kernels use.

TEST INFRASTRUCTURE (CPU only).  It executes the text that sota_imagenet_amd/csrc/asm/dconv_gen.py emits, one workgroup at
a time, so the addressing / layout logic of a hand-scheduled kernel can be checked against numpy before it ever touches a
GPU.  It models values, not time: loads complete at issue.  What it DOES check about the asynchronous parts is the
protocol the kernels rely on (MI355X_MICROARCH.md "Two waves per SIMD" item 7):
  * an LDS byte written by LDS-DMA may be read by the issuing wave only after a covering `s_waitcnt vmcnt(N)`, by any other
    wave only after that wait AND a barrier both have passed;
  * an LDS byte may be overwritten (DMA or ds_write) only in a later barrier epoch than its last read by another wave, and
    a wave must have drained its LDS reads (`lgkmcnt(0)`) before it arrives at a barrier that such a rewrite relies on;
  * a VGPR loaded by ds_read / buffer_load may be consumed only after a wait that covers it.
Violations raise EmuError with the instruction's line.
"""
import re
import struct

import numpy as np

MASK32 = 0xFFFFFFFF


class EmuError(RuntimeError):
    pass


def _f32(u):
    return np.asarray(u, dtype=np.uint32).view(np.float32)


def _u32(f):
    return np.asarray(f, dtype=np.float32).view(np.uint32)


def bf16_round_rne(f):
    """fp32 array -> bf16 bits (uint32 in the low 16 bits), round to nearest even, NaN kept."""
    u = _u32(f).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    nan = np.isnan(np.asarray(f, dtype=np.float32))
    r = np.where(nan, (u >> 16) | 0x40, r)
    return (r & 0xFFFF).astype(np.uint32)


class Memory:
    """Flat device memory: named allocations at fake 48-bit addresses."""

    def __init__(self):
        self.base = 0x7F0000000000
        self.next = self.base
        self.regions = []  # (start, bytearray-backed np.uint8 array)

    def alloc(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        start = self.next
        self.regions.append((start, a))
        self.next = (start + a.size + 0xFFFF) & ~0xFFFF
        self.next += 0x10000  # guard gap
        return start

    def find(self, addr, n):
        for start, a in self.regions:
            if start <= addr and addr + n <= start + a.size:
                return a, addr - start
        raise EmuError("global access outside every allocation: 0x%x (+%d)" % (addr, n))

    def read(self, addr, n):
        a, o = self.find(addr, n)
        return a[o:o + n]

    def write(self, addr, data):
        a, o = self.find(addr, len(data))
        a[o:o + len(data)] = data

    def array(self, start, dtype, shape):
        for s, a in self.regions:
            if s == start:
                return a.view(dtype).reshape(shape)
        raise KeyError(start)


_RE_RANGE = re.compile(r"^([vsa])\[(\d+):(\d+)\]$")
_RE_REG = re.compile(r"^([vsa])(\d+)$")


class Operand:
    __slots__ = ("kind", "idx", "n", "val")

    def __init__(self, kind, idx=0, n=1, val=0):
        self.kind, self.idx, self.n, self.val = kind, idx, n, val


def parse_operand(t):
    t = t.strip()
    m = _RE_RANGE.match(t)
    if m:
        return Operand(m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
    m = _RE_REG.match(t)
    if m:
        return Operand(m.group(1), int(m.group(2)), 1)
    if t in ("m0", "vcc", "exec", "scc", "vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "off"):
        return Operand(t)
    try:
        if t.startswith("0x") or t.startswith("-0x"):
            return Operand("imm", val=int(t, 16) & MASK32)
        if re.match(r"^-?\d+$", t):
            return Operand("imm", val=int(t) & MASK32)
        if re.match(r"^-?\d*\.\d+(e[-+]?\d+)?$", t) or t in ("1.0", "0.5", "2.0", "4.0", "-1.0"):
            return Operand("imm", val=int(_u32(np.float32(float(t)))))
    except ValueError:
        pass
    return Operand("label", val=t)


class Inst:
    __slots__ = ("op", "ops", "mods", "text", "lineno")


def parse_program(text):
    insts, labels = [], {}
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = raw.split(";")[0].split("//")[0].strip()
        if not line or line.startswith("."):
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(insts)
            continue
        parts = line.split(None, 1)
        ins = Inst()
        ins.op = parts[0]
        ins.text = line
        ins.lineno = lineno
        ins.ops, ins.mods = [], {}
        rest = parts[1] if len(parts) > 1 else ""
        if ins.op == "s_waitcnt":
            for m in re.finditer(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", rest):
                ins.mods[m.group(1)] = int(m.group(2))
            insts.append(ins)
            continue
        # split operands at top-level commas; trailing modifiers are space-separated tokens after the last operand
        toks = [t.strip() for t in rest.split(",")] if rest else []
        if toks:
            last = toks[-1].split()
            toks[-1] = last[0] if last else ""
            for md in last[1:]:
                if ":" in md:
                    k, v = md.split(":", 1)
                    ins.mods[k] = int(v, 0) if re.match(r"^-?(0x)?[0-9a-fA-F]+$", v) else v
                else:
                    ins.mods[md] = True
        ins.ops = [parse_operand(t) for t in toks if t != ""]
        insts.append(ins)
    return insts, labels


class Wave:
    def __init__(self, wid, nthreads_in_wg):
        self.wid = wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint32)
        self.m0 = 0
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        # asynchronous-protocol bookkeeping
        self.vm_ops = []      # outstanding vector-memory ops, oldest first: dict(kind, lds slots | vgprs)
        self.lgkm_ops = []    # outstanding LDS / scalar-memory ops: dict(vgprs | sgprs)
        self.pending_v = {}   # vgpr index -> op record still in flight
        self.pending_s = {}
        self.icount = 0


class Emulator:
    LDS_BYTES = 160 * 1024

    def __init__(self, text, mem, lds_bytes=None, check=True, sinks=None, dontcare=None):
        # dontcare: (first byte, end byte) LDS ranges whose READS are exempt from the DMA-protocol checks: the generator declares
        # them for fragment reads that run a few positions past a tile into whatever follows it and only feed discarded lanes
        self.dontcare = dontcare or []
        self.sinks = sinks or []  # (first byte, end byte) LDS regions written by padding DMAs and never read: exempt from the write checks
        self.insts, self.labels = parse_program(text)
        self.mem = mem
        self.lds = np.zeros(lds_bytes or self.LDS_BYTES, dtype=np.uint8)
        self.check = check
        nslot = self.lds.size // 16
        # per 16-byte LDS slot: DMA state
        self.slot_state = np.zeros(nslot, dtype=np.int8)   # 0 clean, 1 in flight, 2 landed (issuer only)
        self.slot_owner = np.full(nslot, -1, dtype=np.int16)
        self.slot_read_epoch = np.full((nslot, 8), -1, dtype=np.int32)  # last barrier epoch in which wave w read the slot
        self.epoch = 0
        self.stats = {}

    # ------------------------------------------------------------------------------------------------------------------
    def run_workgroup(self, nwaves, kernarg_addr, wg_id=(0, 0, 0), user_sgprs=2, max_inst=5_000_000):
        waves = []
        for w in range(nwaves):
            wv = Wave(w, nwaves * 64)
            wv.s[0] = kernarg_addr & MASK32
            wv.s[1] = (kernarg_addr >> 32) & MASK32
            k = user_sgprs
            for i in range(3):
                wv.s[k + i] = wg_id[i]
            wv.v[0] = np.arange(64, dtype=np.uint32) + 64 * w
            waves.append(wv)
        self.waves = waves
        total = 0
        while True:
            progressed = False
            for wv in waves:
                if wv.done or wv.at_barrier:
                    continue
                progressed = True
                while not wv.done and not wv.at_barrier:
                    self.step(wv)
                    total += 1
                    if total > max_inst:
                        raise EmuError("instruction budget exceeded (endless loop?)")
            alive = [w for w in waves if not w.done]
            if not alive:
                break
            if all(w.at_barrier for w in alive):
                if len(alive) != len(waves):
                    raise EmuError("barrier reached after some waves of the workgroup ended")
                self.release_barrier()
            elif not progressed:
                raise EmuError("deadlock")
        return total

    def release_barrier(self):
        self.epoch += 1
        # landed DMA bytes become visible to every wave
        self.slot_state[self.slot_state == 2] = 0
        for wv in self.waves:
            wv.at_barrier = False

    # ------------------------------------------------------------------------------------------------------------------
    def err(self, ins, msg):
        raise EmuError("line %d: %s\n    %s" % (ins.lineno, msg, ins.text))

    # scalar operand read
    def sval(self, wv, o, ins):
        k = o.kind
        if k == "s":
            if self.check and o.idx in wv.pending_s:
                self.err(ins, "s%d read before the s_load that writes it was waited for" % o.idx)
            return int(wv.s[o.idx])
        if k == "imm":
            return o.val
        if k == "m0":
            return wv.m0
        if k == "vcc_lo":
            return wv.vcc & MASK32
        if k == "vcc_hi":
            return wv.vcc >> 32
        if k == "exec_lo":
            return wv.exec & MASK32
        if k == "exec_hi":
            return wv.exec >> 32
        if k == "scc":
            return wv.scc
        self.err(ins, "bad scalar operand kind %s" % k)

    def sval64(self, wv, o, ins):
        if o.kind == "s":
            return int(wv.s[o.idx]) | (int(wv.s[o.idx + 1]) << 32)
        if o.kind == "imm":
            v = o.val
            return v | (0xFFFFFFFF00000000 if v & 0x80000000 else 0)  # 32-bit literals sign-extend for 64-bit operands
        if o.kind == "vcc":
            return wv.vcc
        if o.kind == "exec":
            return wv.exec
        self.err(ins, "bad 64-bit scalar operand")

    def swrite(self, wv, o, val, ins):
        val &= MASK32
        if o.kind == "s":
            wv.s[o.idx] = val
        elif o.kind == "m0":
            wv.m0 = val
        elif o.kind == "vcc_lo":
            wv.vcc = (wv.vcc & ~MASK32) | val
        elif o.kind == "exec_lo":
            wv.exec = (wv.exec & ~MASK32) | val
        elif o.kind == "exec_hi":
            wv.exec = (wv.exec & MASK32) | (val << 32)
        elif o.kind == "vcc_hi":
            wv.vcc = (wv.vcc & MASK32) | (val << 32)
        else:
            self.err(ins, "bad scalar destination")

    def swrite64(self, wv, o, val, ins):
        val &= (1 << 64) - 1
        if o.kind == "s":
            wv.s[o.idx] = val & MASK32
            wv.s[o.idx + 1] = val >> 32
        elif o.kind == "vcc":
            wv.vcc = val
        elif o.kind == "exec":
            wv.exec = val
        else:
            self.err(ins, "bad 64-bit scalar destination")

    # vector operand read -> uint32[64]
    def vval(self, wv, o, ins, sub=0):
        k = o.kind
        if k == "v":
            if self.check and (o.idx + sub) in wv.pending_v:
                self.err(ins, "v%d consumed before the load that writes it was waited for" % (o.idx + sub))
            return wv.v[o.idx + sub]
        if k == "a":
            if self.check and (1000 + o.idx + sub) in wv.pending_v:
                self.err(ins, "a%d consumed before the load that writes it was waited for" % (o.idx + sub))
            return wv.a[o.idx + sub]
        if k in ("s", "imm", "m0", "vcc_lo", "exec_lo", "vcc_hi", "exec_hi"):
            return np.full(64, self.sval(wv, o, ins), dtype=np.uint32)
        self.err(ins, "bad vector operand kind %s" % k)

    _lane_cache = {}

    def bits64(self, e):
        m = self._lane_cache.get(e)
        if m is None:
            m = np.array([(e >> i) & 1 for i in range(64)], dtype=bool)
            if len(self._lane_cache) < 4096:
                self._lane_cache[e] = m
        return m

    def lanes(self, wv):
        return self.bits64(wv.exec)

    def vwrite(self, wv, o, val, ins, sub=0, mask=None):
        if mask is None:
            mask = self.lanes(wv)
        val = np.asarray(val, dtype=np.uint32)
        if o.kind == "v":
            idx = o.idx + sub
            if self.check and idx in wv.pending_v:
                self.err(ins, "v%d overwritten while a load into it is in flight" % idx)
            wv.v[idx] = np.where(mask, val, wv.v[idx])
        elif o.kind == "a":
            if self.check and (1000 + o.idx + sub) in wv.pending_v:
                self.err(ins, "a%d overwritten while a load into it is in flight" % (o.idx + sub))
            wv.a[o.idx + sub] = np.where(mask, val, wv.a[o.idx + sub])
        else:
            self.err(ins, "bad vector destination")

    # ------------------------------------------------------------------------------------------------------------------
    def retire_vm(self, wv, keep):
        while len(wv.vm_ops) > keep:
            op = wv.vm_ops.pop(0)
            if op["kind"] == "dma":
                sl = op["slots"]
                own = (self.slot_owner[sl] == wv.wid) & (self.slot_state[sl] == 1) & (self.slot_seq[sl] == op["seq"])
                self.slot_state[sl[own]] = 2
            elif op["kind"] == "load":
                for r in op["vgprs"]:
                    if wv.pending_v.get(r) is op:
                        del wv.pending_v[r]

    def retire_lgkm(self, wv, keep):
        # LDS returns in order; scalar loads may return out of order, so a counted wait with SMEM outstanding retires nothing
        if keep > 0 and any(op["kind"] == "smem" for op in wv.lgkm_ops):
            return
        while len(wv.lgkm_ops) > keep:
            op = wv.lgkm_ops.pop(0)
            for r in op.get("vgprs", ()):
                if wv.pending_v.get(r) is op:
                    del wv.pending_v[r]
            for r in op.get("sgprs", ()):
                if wv.pending_s.get(r) is op:
                    del wv.pending_s[r]

    # LDS access checks
    def lds_read_check(self, wv, addrs, nbytes, mask, ins):
        if not self.check:
            return
        sl = (addrs[mask] // 16).astype(np.int64)  # accesses are naturally aligned and <= 16 bytes: one slot per lane
        for lo, hi in self.dontcare:
            sl = sl[(sl < lo // 16) | (sl >= hi // 16)]
        if sl.size == 0:
            return
        st = self.slot_state[sl]
        if (st == 1).any():
            i = int(np.nonzero(st == 1)[0][0])
            self.err(ins, "ds_read of LDS bytes 0x%x whose LDS-DMA (wave %d) has not been waited for" % (int(sl[i]) * 16, self.slot_owner[sl[i]]))
        bad = (st == 2) & (self.slot_owner[sl] != wv.wid)
        if bad.any():
            i = int(np.nonzero(bad)[0][0])
            self.err(ins, "ds_read of LDS bytes 0x%x landed by wave %d's DMA but not yet published by a barrier" % (int(sl[i]) * 16, self.slot_owner[sl[i]]))
        self.slot_read_epoch[sl, wv.wid] = self.epoch

    def lds_write_check(self, wv, slots, ins):
        if not self.check:
            return
        for lo, hi in self.sinks:
            slots = slots[(slots < lo // 16) | (slots >= hi // 16)]
        if slots.size == 0:
            return
        re_ = self.slot_read_epoch[slots]          # [n, 8]
        others = np.ones(8, dtype=bool)
        others[wv.wid] = False
        bad = (re_[:, others] >= self.epoch).any()
        if bad:
            self.err(ins, "LDS bytes rewritten in the barrier epoch in which another wave read them (WAR race)")
        if (self.slot_state[slots] == 1).any():
            self.err(ins, "LDS bytes rewritten while an LDS-DMA into them is in flight")

    # ------------------------------------------------------------------------------------------------------------------
    slot_seq = None

    def step(self, wv):
        if wv.pc >= len(self.insts):
            raise EmuError("pc ran off the program")
        ins = self.insts[wv.pc]
        wv.pc += 1
        wv.icount += 1
        self.stats[ins.op] = self.stats.get(ins.op, 0) + 1
        op = ins.op
        fn = getattr(self, "i_" + op.replace(".", "_"), None)
        if fn is None:
            base = op
            for suf in ("_e32", "_e64", "_dpp"):
                if base.endswith(suf):
                    base = base[: -len(suf)]
            fn = getattr(self, "i_" + base, None)
        if fn is None:
            self.err(ins, "unsupported instruction")
        fn(wv, ins)

    # ---- scalar ------------------------------------------------------------------------------------------------------
    def i_s_nop(self, wv, ins):
        pass

    i_s_sleep = i_s_nop
    i_s_setprio = i_s_nop
    i_s_sethalt = i_s_nop

    def i_s_endpgm(self, wv, ins):
        if self.check and (wv.vm_ops and any(o["kind"] == "dma" for o in wv.vm_ops)):
            pass
        wv.done = True

    def i_s_barrier(self, wv, ins):
        if self.check and any(op["kind"] == "lds" for op in wv.lgkm_ops):
            self.err(ins, "s_barrier with LDS reads still outstanding (lgkmcnt not drained): a rewrite after the barrier could race them")
        wv.at_barrier = True

    def i_s_waitcnt(self, wv, ins):
        if "vmcnt" in ins.mods:
            self.retire_vm(wv, ins.mods["vmcnt"])
        if "lgkmcnt" in ins.mods:
            self.retire_lgkm(wv, ins.mods["lgkmcnt"])

    def i_s_mov_b32(self, wv, ins):
        self.swrite(wv, ins.ops[0], self.sval(wv, ins.ops[1], ins), ins)

    def i_s_mov_b64(self, wv, ins):
        self.swrite64(wv, ins.ops[0], self.sval64(wv, ins.ops[1], ins), ins)

    def _s2(self, wv, ins):
        return self.sval(wv, ins.ops[1], ins), self.sval(wv, ins.ops[2], ins)

    def i_s_add_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a + b
        wv.scc = 1 if r > MASK32 else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_add_i32(self, wv, ins):
        a, b = self._s2(wv, ins)
        self.swrite(wv, ins.ops[0], a + b, ins)
        wv.scc = 0

    def i_s_addc_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a + b + wv.scc
        wv.scc = 1 if r > MASK32 else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_sub_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        wv.scc = 1 if b > a else 0
        self.swrite(wv, ins.ops[0], a - b, ins)

    i_s_sub_i32 = i_s_sub_u32

    def i_s_subb_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a - b - wv.scc
        wv.scc = 1 if r < 0 else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_mul_i32(self, wv, ins):
        a, b = self._s2(wv, ins)
        self.swrite(wv, ins.ops[0], a * b, ins)

    def i_s_mul_hi_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        self.swrite(wv, ins.ops[0], (a * b) >> 32, ins)

    def i_s_lshl_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = (a << (b & 31)) & MASK32
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_lshr_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a >> (b & 31)
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_and_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a & b
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_or_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a | b
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_xor_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        r = a ^ b
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def i_s_and_b64(self, wv, ins):
        r = self.sval64(wv, ins.ops[1], ins) & self.sval64(wv, ins.ops[2], ins)
        wv.scc = 1 if r else 0
        self.swrite64(wv, ins.ops[0], r, ins)

    def i_s_cmp_eq_u64(self, wv, ins):
        wv.scc = 1 if self.sval64(wv, ins.ops[0], ins) == self.sval64(wv, ins.ops[1], ins) else 0

    def i_s_andn2_b64(self, wv, ins):
        r = self.sval64(wv, ins.ops[1], ins) & ~self.sval64(wv, ins.ops[2], ins) & ((1 << 64) - 1)
        wv.scc = 1 if r else 0
        self.swrite64(wv, ins.ops[0], r, ins)

    def i_s_min_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        wv.scc = 1 if a <= b else 0
        self.swrite(wv, ins.ops[0], min(a, b), ins)

    def i_s_cselect_b32(self, wv, ins):
        a, b = self._s2(wv, ins)
        self.swrite(wv, ins.ops[0], a if wv.scc else b, ins)

    def i_s_cselect_b64(self, wv, ins):
        a, b = self.sval64(wv, ins.ops[1], ins), self.sval64(wv, ins.ops[2], ins)
        self.swrite64(wv, ins.ops[0], a if wv.scc else b, ins)

    def i_s_bfe_u32(self, wv, ins):
        a, b = self._s2(wv, ins)
        off, width = b & 31, (b >> 16) & 0x7F
        r = (a >> off) & ((1 << width) - 1)
        wv.scc = 1 if r else 0
        self.swrite(wv, ins.ops[0], r, ins)

    def _scmp(self, wv, ins, f):
        a = self.sval(wv, ins.ops[0], ins)
        b = self.sval(wv, ins.ops[1], ins)
        wv.scc = 1 if f(a, b) else 0

    def i_s_cmp_eq_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a == b)

    i_s_cmp_eq_i32 = i_s_cmp_eq_u32

    def i_s_cmp_lg_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a != b)

    i_s_cmp_lg_i32 = i_s_cmp_lg_u32

    def i_s_cmp_lt_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a < b)

    def i_s_cmp_ge_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a >= b)

    def i_s_cmp_gt_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a > b)

    def i_s_cmp_le_u32(self, wv, ins):
        self._scmp(wv, ins, lambda a, b: a <= b)

    def _branch(self, wv, ins, cond):
        if cond:
            lab = ins.ops[0].val
            if lab not in self.labels:
                self.err(ins, "unknown label")
            wv.pc = self.labels[lab]

    def i_s_branch(self, wv, ins):
        self._branch(wv, ins, True)

    def i_s_cbranch_scc0(self, wv, ins):
        self._branch(wv, ins, wv.scc == 0)

    def i_s_cbranch_scc1(self, wv, ins):
        self._branch(wv, ins, wv.scc == 1)

    def i_s_cbranch_vccz(self, wv, ins):
        self._branch(wv, ins, wv.vcc == 0)

    def i_s_cbranch_vccnz(self, wv, ins):
        self._branch(wv, ins, wv.vcc != 0)

    def i_s_cbranch_execz(self, wv, ins):
        self._branch(wv, ins, wv.exec == 0)

    def _s_load(self, wv, ins, n):
        dst, base = ins.ops[0], ins.ops[1]
        off = self.sval(wv, ins.ops[2], ins) if len(ins.ops) > 2 else 0
        addr = self.sval64(wv, base, ins) + off
        raw = self.mem.read(addr, 4 * n).view(np.uint32)
        rec = {"kind": "smem", "sgprs": list(range(dst.idx, dst.idx + n))}
        for i in range(n):
            wv.s[dst.idx + i] = raw[i]
            wv.pending_s[dst.idx + i] = rec
        wv.lgkm_ops.append(rec)

    def i_s_load_dword(self, wv, ins):
        self._s_load(wv, ins, 1)

    def i_s_load_dwordx2(self, wv, ins):
        self._s_load(wv, ins, 2)

    def i_s_load_dwordx4(self, wv, ins):
        self._s_load(wv, ins, 4)

    def i_s_load_dwordx8(self, wv, ins):
        self._s_load(wv, ins, 8)

    def i_s_load_dwordx16(self, wv, ins):
        self._s_load(wv, ins, 16)

    # ---- vector integer ----------------------------------------------------------------------------------------------
    def _v2(self, wv, ins):
        return self.vval(wv, ins.ops[1], ins).astype(np.uint64), self.vval(wv, ins.ops[2], ins).astype(np.uint64)

    def _v3(self, wv, ins):
        return (self.vval(wv, ins.ops[1], ins).astype(np.uint64), self.vval(wv, ins.ops[2], ins).astype(np.uint64),
                self.vval(wv, ins.ops[3], ins).astype(np.uint64))

    def _vw(self, wv, ins, r):
        self.vwrite(wv, ins.ops[0], (np.asarray(r).astype(np.uint64) & MASK32).astype(np.uint32), ins)

    def i_v_mov_b32(self, wv, ins):
        src = self.vval(wv, ins.ops[1], ins)
        mask = None
        if "row_shr" in ins.mods or "row_shl" in ins.mods:
            src, valid = self._dpp(ins, src)
            src = np.where(valid, src, 0 if ins.mods.get("bound_ctrl") else wv.v[ins.ops[0].idx])
        elif "row_ror" in ins.mods:
            # rotate right within each row of 16 lanes: lane i receives lane (i - n) mod 16 of its row
            n = int(ins.mods["row_ror"])
            lane = np.arange(64)
            src = src[(lane & ~15) | ((lane - n) & 15)]
        if "row_ror" in ins.mods or "row_shr" in ins.mods or "row_shl" in ins.mods:
            # bank_mask bit b enables lanes 4b .. 4b + 3 of every row, row_mask bit r enables row r (destination write only)
            lane = np.arange(64)
            bm, rm = int(ins.mods.get("bank_mask", 0xF)), int(ins.mods.get("row_mask", 0xF))
            mask = self.lanes(wv) & (((bm >> ((lane & 15) >> 2)) & 1) == 1) & (((rm >> (lane >> 4)) & 1) == 1)
        self.vwrite(wv, ins.ops[0], src, ins, mask=mask)

    def i_v_add_u32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a + b)

    def i_v_sub_u32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a - b + (1 << 32))

    def i_v_subrev_u32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, b - a + (1 << 32))

    def i_v_mul_lo_u32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a * b)

    def i_v_mul_hi_u32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, (a.astype(np.uint64) * b.astype(np.uint64)) >> np.uint64(32))

    def i_v_mul_u32_u24(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, (a & 0xFFFFFF) * (b & 0xFFFFFF))

    def i_v_mad_u32_u24(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a & 0xFFFFFF) * (b & 0xFFFFFF) + c)

    def i_v_lshlrev_b32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, b << (a & 31))

    def i_v_lshrrev_b32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, b >> (a & 31))

    def i_v_and_b32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a & b)

    def i_v_or_b32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a | b)

    def i_v_xor_b32(self, wv, ins):
        a, b = self._v2(wv, ins)
        self._vw(wv, ins, a ^ b)

    def i_v_lshl_add_u32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a << (b & 31)) + c)

    def i_v_add_lshl_u32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a + b) << (c & 31))

    def i_v_lshl_or_b32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a << (b & 31)) | c)

    def i_v_bfi_b32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a & b) | ((a ^ 0xFFFFFFFF) & c))

    def i_v_and_or_b32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a & b) | c)

    def i_v_add3_u32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, a + b + c)

    def i_v_bfe_u32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        self._vw(wv, ins, (a >> (b & 31)) & ((1 << (c & 31)) - 1))

    def i_v_bfe_i32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        a, b, w = a.astype(np.int64), b.astype(np.int64), (c & 31).astype(np.int64)
        f = (a >> (b & 31)) & ((np.int64(1) << w) - 1)
        sign = (f >> (w - 1)) & 1
        self._vw(wv, ins, np.where(sign == 1, f - (np.int64(1) << w) + (np.int64(1) << 32), f).astype(np.uint64))

    def i_v_perm_b32(self, wv, ins):
        a, b, c = self._v3(wv, ins)
        src = (a << 32) | b   # bytes 0-3 from src1 (b), 4-7 from src0 (a)
        r = np.zeros(64, dtype=np.uint64)
        for i in range(4):
            sel = (c >> (8 * i)) & 0xFF
            byte = np.where(sel < 8, (src >> (8 * (sel & 7))) & 0xFF, np.where(sel == 12, 0, 0xFF))
            r |= byte << (8 * i)
        self._vw(wv, ins, r)

    def _vcmp(self, wv, ins, f):
        # v_cmp_xx_u32 vcc|s[..], a, b
        a = self.vval(wv, ins.ops[1], ins).astype(np.int64)
        b = self.vval(wv, ins.ops[2], ins).astype(np.int64)
        res = f(a, b) & self.lanes(wv)
        val = 0
        for l in np.nonzero(res)[0]:
            val |= 1 << int(l)
        self.swrite64(wv, ins.ops[0], val, ins)

    def _vcmpf(self, wv, ins, f):
        # v_cmp_xx_f32 vcc|s[..], a, b
        a = _f32(self.vval(wv, ins.ops[1], ins))
        b = _f32(self.vval(wv, ins.ops[2], ins))
        with np.errstate(all="ignore"):
            res = f(a, b) & self.lanes(wv)
        val = 0
        for l in np.nonzero(res)[0]:
            val |= 1 << int(l)
        self.swrite64(wv, ins.ops[0], val, ins)

    def i_v_cmp_lt_f32(self, wv, ins):
        self._vcmpf(wv, ins, lambda a, b: a < b)

    def i_v_cmp_gt_f32(self, wv, ins):
        self._vcmpf(wv, ins, lambda a, b: a > b)

    def i_v_addc_co_u32(self, wv, ins):
        # v_addc_co_u32 dst, carry_out (vcc | s[..]), a, b, carry_in (vcc | s[..])
        a = self.vval(wv, ins.ops[2], ins).astype(np.int64)
        b = self.vval(wv, ins.ops[3], ins).astype(np.int64)
        cin = self.bits64(self.sval64(wv, ins.ops[4], ins)).astype(np.int64)
        r = a + b + cin
        act = self.lanes(wv)
        val = 0
        for l in np.nonzero((r > MASK32) & act)[0]:
            val |= 1 << int(l)
        self.vwrite(wv, ins.ops[0], (r & MASK32).astype(np.uint32), ins)
        self.swrite64(wv, ins.ops[1], val, ins)

    def i_v_pk_max_i16(self, wv, ins):
        a = self.vval(wv, ins.ops[1], ins).astype(np.int64)
        b = self.vval(wv, ins.ops[2], ins).astype(np.int64)
        s16 = lambda x: np.where(x & 0x8000, x - 0x10000, x)
        lo = np.maximum(s16(a & 0xFFFF), s16(b & 0xFFFF)) & 0xFFFF
        hi = np.maximum(s16((a >> 16) & 0xFFFF), s16((b >> 16) & 0xFFFF)) & 0xFFFF
        self.vwrite(wv, ins.ops[0], ((hi << 16) | lo).astype(np.uint32), ins)

    def i_v_pk_min_u16(self, wv, ins):
        a = self.vval(wv, ins.ops[1], ins).astype(np.int64)
        b = self.vval(wv, ins.ops[2], ins).astype(np.int64)
        lo = np.minimum(a & 0xFFFF, b & 0xFFFF)
        hi = np.minimum((a >> 16) & 0xFFFF, (b >> 16) & 0xFFFF)
        self.vwrite(wv, ins.ops[0], ((hi << 16) | lo).astype(np.uint32), ins)

    def i_v_cmp_lt_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a < b)

    def i_v_cmp_le_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a <= b)

    def i_v_cmp_gt_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a > b)

    def i_v_cmp_ge_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a >= b)

    def i_v_cmp_eq_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a == b)

    def i_v_cmp_ne_u32(self, wv, ins):
        self._vcmp(wv, ins, lambda a, b: a != b)

    def i_v_cndmask_b32(self, wv, ins):
        # dst, src0 (mask bit 0), src1 (mask bit 1), mask (vcc or s[..])
        a = self.vval(wv, ins.ops[1], ins)
        b = self.vval(wv, ins.ops[2], ins)
        m = self.sval64(wv, ins.ops[3], ins) if len(ins.ops) > 3 else wv.vcc
        sel = self.bits64(m)
        self.vwrite(wv, ins.ops[0], np.where(sel, b, a), ins)

    def i_v_readfirstlane_b32(self, wv, ins):
        src = self.vval(wv, ins.ops[1], ins)
        e = wv.exec
        lane = 0
        if e:
            while not (e >> lane) & 1:
                lane += 1
        self.swrite(wv, ins.ops[0], int(src[lane]), ins)

    def i_v_readlane_b32(self, wv, ins):
        src = self.vval(wv, ins.ops[1], ins)
        lane = self.sval(wv, ins.ops[2], ins) & 63
        self.swrite(wv, ins.ops[0], int(src[lane]), ins)

    def i_v_accvgpr_read_b32(self, wv, ins):
        self.vwrite(wv, ins.ops[0], self.vval(wv, ins.ops[1], ins), ins)

    def i_v_accvgpr_write_b32(self, wv, ins):
        self.vwrite(wv, ins.ops[0], self.vval(wv, ins.ops[1], ins), ins)

    # ---- vector float --------------------------------------------------------------------------------------------------
    def _dpp(self, ins, src):
        """returns (permuted source, valid mask) for row_shr:n / row_shl:n within 16-lane rows"""
        lane = np.arange(64)
        if "row_shr" in ins.mods:
            n = int(ins.mods["row_shr"])
            srcl = lane - n
            valid = (lane % 16) >= n
        else:
            n = int(ins.mods["row_shl"])
            srcl = lane + n
            valid = (lane % 16) + n < 16
        srcl = np.clip(srcl, 0, 63)
        return src[srcl], valid

    def _fbin(self, wv, ins, f):
        a = self.vval(wv, ins.ops[1], ins)
        b = self.vval(wv, ins.ops[2], ins)
        mask = self.lanes(wv)
        if "row_shr" in ins.mods or "row_shl" in ins.mods:
            a, valid = self._dpp(ins, a)
            if ins.mods.get("bound_ctrl"):
                a = np.where(valid, a, 0)
            else:
                mask = mask & valid
        with np.errstate(all="ignore"):
            r = f(_f32(a), _f32(b)).astype(np.float32)
        self.vwrite(wv, ins.ops[0], _u32(r), ins, mask=mask)

    def i_v_add_f32(self, wv, ins):
        self._fbin(wv, ins, lambda a, b: a + b)

    def i_v_sub_f32(self, wv, ins):
        self._fbin(wv, ins, lambda a, b: a - b)

    def i_v_mul_f32(self, wv, ins):
        self._fbin(wv, ins, lambda a, b: a * b)

    def i_v_max_f32(self, wv, ins):
        self._fbin(wv, ins, np.maximum)

    def i_v_fma_f32(self, wv, ins):
        a = _f32(self.vval(wv, ins.ops[1], ins)).astype(np.float64)
        b = _f32(self.vval(wv, ins.ops[2], ins)).astype(np.float64)
        c = _f32(self.vval(wv, ins.ops[3], ins)).astype(np.float64)
        with np.errstate(all="ignore"):
            r = (a * b + c).astype(np.float32)  # a*b exact in fp64 for fp32 inputs; one rounding (double rounding is negligible here)
        self.vwrite(wv, ins.ops[0], _u32(r), ins)

    i_v_fmac_f32 = None  # (not used: keeps three-operand forms explicit)

    def i_v_cvt_pk_bf16_f32(self, wv, ins):
        lo = bf16_round_rne(_f32(self.vval(wv, ins.ops[1], ins)))
        hi = bf16_round_rne(_f32(self.vval(wv, ins.ops[2], ins)))
        self.vwrite(wv, ins.ops[0], lo | (hi << 16), ins)

    def i_v_cvt_f32_u32(self, wv, ins):
        self.vwrite(wv, ins.ops[0], _u32(self.vval(wv, ins.ops[1], ins).astype(np.float32)), ins)

    def i_v_cvt_f32_ubyte0(self, wv, ins):
        self.vwrite(wv, ins.ops[0], _u32((self.vval(wv, ins.ops[1], ins) & 0xFF).astype(np.float32)), ins)

    # ---- MFMA ------------------------------------------------------------------------------------------------------------
    def i_v_mfma_f32_16x16x32_bf16(self, wv, ins):
        d, a, b, c = ins.ops
        A = np.zeros((16, 32), dtype=np.float64)
        B = np.zeros((32, 16), dtype=np.float64)
        lane = np.arange(64)
        for r in range(4):
            av = self.vval(wv, a, ins, r)
            bv = self.vval(wv, b, ins, r)
            for h in range(2):
                k = 8 * (lane >> 4) + 2 * r + h
                A[lane & 15, k] = _f32(((av >> (16 * h)) & 0xFFFF) << 16)
                B[k, lane & 15] = _f32(((bv >> (16 * h)) & 0xFFFF) << 16)
        C = np.zeros((16, 16), dtype=np.float64)
        for r in range(4):
            cv = self.vval(wv, c, ins, r) if c.kind != "imm" else np.full(64, c.val, dtype=np.uint32)
            C[4 * (lane >> 4) + r, lane & 15] = _f32(cv)
        D = (A @ B + C).astype(np.float32)
        full = np.ones(64, dtype=bool)
        for r in range(4):
            self.vwrite(wv, d, _u32(D[4 * (lane >> 4) + r, lane & 15]), ins, r, mask=full)

    def i_v_rcp_f32(self, wv, ins):
        a = _f32(self.vval(wv, ins.ops[1], ins)).astype(np.float64)
        with np.errstate(all="ignore"):
            self.vwrite(wv, ins.ops[0], _u32((1.0 / a).astype(np.float32)), ins)

    @staticmethod
    def _e4m3(b):
        """OCP e4m3fn bytes -> float64 (no infinities; 0x7f / 0xff are NaN)"""
        b = b.astype(np.int64)
        sgn = np.where(b & 0x80, -1.0, 1.0)
        ex, man = (b >> 3) & 0xF, b & 7
        val = np.where(ex == 0, man / 8.0 * 2.0 ** -6, (1 + man / 8.0) * 2.0 ** (ex - 7.0))
        val = np.where((ex == 15) & (man == 7), np.nan, val)
        return sgn * val

    def i_v_mfma_f32_16x16x128_f8f6f4(self, wv, ins):
        """D = A (16 x 128) * B (128 x 16) + C on e4m3 bytes (cbsz = blgp = 0); lane l supplies row / column l & 15, k = 32 (l >> 4) + 4 r + byte of register r"""
        d, a, b, c = ins.ops
        if any(k in ins.mods for k in ("cbsz", "blgp")):
            self.err(ins, "only the e4m3 x e4m3 form (cbsz = blgp = 0) is modelled")
        A = np.zeros((16, 128), dtype=np.float64)
        B = np.zeros((128, 16), dtype=np.float64)
        lane = np.arange(64)
        for r in range(8):
            av = self.vval(wv, a, ins, r)
            bv = self.vval(wv, b, ins, r)
            for h in range(4):
                k = 32 * (lane >> 4) + 4 * r + h
                A[lane & 15, k] = self._e4m3((av >> (8 * h)) & 0xFF)
                B[k, lane & 15] = self._e4m3((bv >> (8 * h)) & 0xFF)
        C = np.zeros((16, 16), dtype=np.float64)
        for r in range(4):
            cv = self.vval(wv, c, ins, r) if c.kind != "imm" else np.full(64, c.val, dtype=np.uint32)
            C[4 * (lane >> 4) + r, lane & 15] = _f32(cv)
        D = (A @ B + C).astype(np.float32)
        full = np.ones(64, dtype=bool)
        for r in range(4):
            self.vwrite(wv, d, _u32(D[4 * (lane >> 4) + r, lane & 15]), ins, r, mask=full)

    # ---- LDS ---------------------------------------------------------------------------------------------------------------
    def _ds_addr(self, wv, ins, o):
        return (self.vval(wv, o, ins).astype(np.int64) + int(ins.mods.get("offset", 0))) & MASK32

    def _ds_read(self, wv, ins, nbytes):
        dst, addr_o = ins.ops[0], ins.ops[1]
        addrs = self._ds_addr(wv, ins, addr_o)
        mask = self.lanes(wv)
        if (addrs[mask] + nbytes > self.lds.size).any():
            self.err(ins, "ds_read beyond the LDS allocation")
        if ((addrs[mask] % nbytes) != 0).any():
            self.err(ins, "misaligned ds_read")
        self.lds_read_check(wv, addrs, nbytes, mask, ins)
        nd = nbytes // 4
        rec = {"kind": "lds", "vgprs": list(range(dst.idx, dst.idx + nd))}
        lds32 = self.lds.view(np.uint32)
        a4 = np.where(mask, addrs // 4, 0).astype(np.int64)
        for i in range(nd):
            wv.v[dst.idx + i] = np.where(mask, lds32[a4 + i], wv.v[dst.idx + i])
        for r in rec["vgprs"]:
            if self.check and r in wv.pending_v:
                self.err(ins, "v%d is the destination of two loads in flight" % r)
            wv.pending_v[r] = rec
        wv.lgkm_ops.append(rec)

    def i_ds_read_b128(self, wv, ins):
        self._ds_read(wv, ins, 16)

    def i_ds_read_b64_tr_b16(self, wv, ins):
        """hardware transpose read (guide T10): per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p of
        the group supplies the address of row q, columns 4p .. 4p + 3; lane i receives column i of the 4 rows, row q in element q"""
        dst, addr_o = ins.ops[0], ins.ops[1]
        addrs = self._ds_addr(wv, ins, addr_o)
        if wv.exec != (1 << 64) - 1:
            self.err(ins, "ds_read_b64_tr_b16 with partial EXEC")
        if (addrs + 8 > self.lds.size).any():
            self.err(ins, "ds_read beyond the LDS allocation")
        if ((addrs % 8) != 0).any():
            self.err(ins, "misaligned ds_read_b64_tr_b16")
        self.lds_read_check(wv, addrs, 8, self.lanes(wv), ins)
        lds16 = self.lds.view(np.uint16)
        lane = np.arange(64)
        grp, i = lane & ~15, lane & 15
        out = []
        for q in range(4):
            src_lane = grp + 4 * q + (i >> 2)
            out.append(lds16[(addrs[src_lane] + 2 * (i & 3)) // 2].astype(np.uint32))
        wv.v[dst.idx] = out[0] | (out[1] << 16)
        wv.v[dst.idx + 1] = out[2] | (out[3] << 16)
        rec = {"kind": "lds", "vgprs": [dst.idx, dst.idx + 1]}
        for r in rec["vgprs"]:
            if self.check and r in wv.pending_v:
                self.err(ins, "v%d is the destination of two loads in flight" % r)
            wv.pending_v[r] = rec
        wv.lgkm_ops.append(rec)

    def i_ds_bpermute_b32(self, wv, ins):
        """backward permute through the LDS crossbar (no LDS memory): lane i receives src of lane ((addr[i] + offset) / 4) % 64"""
        dst, addr_o, src_o = ins.ops
        addrs = self._ds_addr(wv, ins, addr_o)
        src = self.vval(wv, src_o, ins)
        val = src[(addrs // 4) % 64]
        mask = self.lanes(wv)
        wv.v[dst.idx] = np.where(mask, val, wv.v[dst.idx])
        rec = {"kind": "lds", "vgprs": [dst.idx]}
        if self.check and dst.idx in wv.pending_v:
            self.err(ins, "v%d is the destination of two loads in flight" % dst.idx)
        wv.pending_v[dst.idx] = rec
        wv.lgkm_ops.append(rec)

    def i_ds_read_b64(self, wv, ins):
        self._ds_read(wv, ins, 8)

    def i_ds_read_b32(self, wv, ins):
        self._ds_read(wv, ins, 4)

    def _ds_write(self, wv, ins, nbytes):
        addr_o, data = ins.ops[0], ins.ops[1]
        addrs = self._ds_addr(wv, ins, addr_o)
        mask = self.lanes(wv)
        if (addrs[mask] + nbytes > self.lds.size).any():
            self.err(ins, "ds_write beyond the LDS allocation")
        nd = nbytes // 4
        vals = [self.vval(wv, data, ins, i) for i in range(nd)]
        slots = []
        for l in np.nonzero(mask)[0]:
            a = int(addrs[l])
            slots.extend(range(a // 16, (a + nbytes - 1) // 16 + 1))
        if slots:
            self.lds_write_check(wv, np.unique(np.array(slots)), ins)
        for l in np.nonzero(mask)[0]:
            a = int(addrs[l])
            self.lds[a:a + nbytes] = np.array([vals[i][l] for i in range(nd)], dtype=np.uint32).view(np.uint8)
        wv.lgkm_ops.append({"kind": "ldsw"})

    def i_ds_add_f32(self, wv, ins):
        addr_o, data = ins.ops[0], ins.ops[1]
        addrs = self._ds_addr(wv, ins, addr_o)
        mask = self.lanes(wv)
        if (addrs[mask] + 4 > self.lds.size).any() or ((addrs[mask] % 4) != 0).any():
            self.err(ins, "ds_add_f32 beyond the LDS allocation / misaligned")
        vals = _f32(self.vval(wv, data, ins))
        lds32 = self.lds.view(np.float32)
        for l in np.nonzero(mask)[0]:
            lds32[int(addrs[l]) // 4] = np.float32(lds32[int(addrs[l]) // 4] + vals[l])
        wv.lgkm_ops.append({"kind": "ldsw"})

    def i_ds_write_b128(self, wv, ins):
        self._ds_write(wv, ins, 16)

    def i_ds_write_b64(self, wv, ins):
        self._ds_write(wv, ins, 8)

    def i_ds_write_b32(self, wv, ins):
        self._ds_write(wv, ins, 4)

    # ---- buffer -------------------------------------------------------------------------------------------------------------
    def _buf_addr(self, wv, ins, vaddr_o, srd_o, soff_o):
        srd = [int(wv.s[srd_o.idx + i]) for i in range(4)]
        for i in range(4):
            if self.check and (srd_o.idx + i) in wv.pending_s:
                self.err(ins, "buffer descriptor read before its s_load was waited for")
        base = srd[0] | ((srd[1] & 0xFFFF) << 32)
        stride = (srd[1] >> 16) & 0x3FFF
        if stride != 0:
            self.err(ins, "only raw (stride 0) buffers are modelled")
        nrec = srd[2]
        voff = self.vval(wv, vaddr_o, ins).astype(np.int64) if ins.mods.get("offen") else np.zeros(64, dtype=np.int64)
        ioff = int(ins.mods.get("offset", 0))
        soff = self.sval(wv, soff_o, ins)
        off = voff + ioff
        return base, off, soff, nrec

    def _buf_load(self, wv, ins, nbytes):
        if ins.mods.get("lds"):
            vaddr_o, srd_o, soff_o = ins.ops[0], ins.ops[1], ins.ops[2]
            base, off, soff, nrec = self._buf_addr(wv, ins, vaddr_o, srd_o, soff_o)
            if wv.exec != (1 << 64) - 1:
                self.err(ins, "LDS-DMA with partial EXEC")
            dst0 = (wv.m0 + 0) & 0x3FFFF
            if dst0 % 16:
                self.err(ins, "LDS-DMA destination not 16-byte aligned")
            if dst0 + 64 * nbytes > self.lds.size:
                self.err(ins, "LDS-DMA beyond the LDS allocation")
            slots = np.arange(dst0 // 16, dst0 // 16 + 64 * nbytes // 16)
            in_sink = any(lo <= dst0 and dst0 + 64 * nbytes <= hi for lo, hi in self.sinks)
            dup = self.check and not in_sink and (self.slot_state[slots] == 1).all() and (self.slot_owner[slots] == wv.wid).all()
            old = self.lds[dst0: dst0 + 64 * nbytes].copy() if dup else None
            if not dup:
                self.lds_write_check(wv, slots, ins)
            if self.slot_seq is None:
                self.slot_seq = np.zeros(self.slot_state.size, dtype=np.int64)
            self.dma_seq = getattr(self, "dma_seq", 0) + 1
            for l in range(64):
                o = int(off[l])
                # raw buffer range check: the whole element must lie inside num_records (soffset is not part of the check,
                # it is subtracted from the limit)
                if o + nbytes > nrec or o < 0:
                    data = np.zeros(nbytes, dtype=np.uint8)
                else:
                    if o + soff + nbytes > nrec:
                        self.err(ins, "in-range offset pushed past num_records by soffset (lane %d): the hardware range check is not relied on for this" % l)
                    data = self.mem.read(base + o + soff, nbytes)
                self.lds[dst0 + l * nbytes: dst0 + (l + 1) * nbytes] = data
            if dup and not (old == self.lds[dst0: dst0 + 64 * nbytes]).all():
                self.err(ins, "LDS bytes rewritten with DIFFERENT data while this wave's earlier LDS-DMA into them is in flight")
            self.slot_state[slots] = 1
            self.slot_owner[slots] = wv.wid
            self.slot_seq[slots] = self.dma_seq
            wv.vm_ops.append({"kind": "dma", "slots": slots, "seq": self.dma_seq})
            return
        dst, vaddr_o, srd_o, soff_o = ins.ops
        base, off, soff, nrec = self._buf_addr(wv, ins, vaddr_o, srd_o, soff_o)
        mask = self.lanes(wv)
        nd = max(1, nbytes // 4)
        # (a load may target accumulation registers: they are tracked as registers 1000 + index)
        bank, koff = (wv.a, 1000) if dst.kind == "a" else (wv.v, 0)
        rec = {"kind": "load", "vgprs": list(range(koff + dst.idx, koff + dst.idx + nd))}
        for l in np.nonzero(mask)[0]:
            o = int(off[l])
            if o + nbytes > nrec - soff or o < 0:
                raw = np.zeros(nd, dtype=np.uint32)
            else:
                b = self.mem.read(base + o + soff, nbytes)
                if nbytes >= 4:
                    raw = b.view(np.uint32)
                else:
                    raw = np.array([int.from_bytes(bytes(b), "little")], dtype=np.uint32)
            for i in range(nd):
                bank[dst.idx + i, l] = raw[i]
        for r in rec["vgprs"]:
            if self.check and r in wv.pending_v:
                self.err(ins, "register %d is the destination of two loads in flight" % r)
            wv.pending_v[r] = rec
        wv.vm_ops.append(rec)

    def i_buffer_load_dwordx4(self, wv, ins):
        self._buf_load(wv, ins, 16)

    def i_buffer_load_dwordx2(self, wv, ins):
        self._buf_load(wv, ins, 8)

    def i_buffer_load_dword(self, wv, ins):
        self._buf_load(wv, ins, 4)

    def i_buffer_load_ubyte(self, wv, ins):
        self._buf_load(wv, ins, 1)

    def i_buffer_load_ushort(self, wv, ins):
        self._buf_load(wv, ins, 2)

    def _buf_store(self, wv, ins, nbytes):
        data, vaddr_o, srd_o, soff_o = ins.ops
        base, off, soff, nrec = self._buf_addr(wv, ins, vaddr_o, srd_o, soff_o)
        mask = self.lanes(wv)
        nd = nbytes // 4
        vals = [self.vval(wv, data, ins, i) for i in range(nd)]
        for l in np.nonzero(mask)[0]:
            o = int(off[l])
            if o + nbytes > nrec - soff or o < 0:
                continue
            self.mem.write(base + o + soff, np.array([vals[i][l] for i in range(nd)], dtype=np.uint32).view(np.uint8))
        wv.vm_ops.append({"kind": "store"})

    def i_buffer_store_dwordx4(self, wv, ins):
        self._buf_store(wv, ins, 16)

    def i_buffer_store_dwordx2(self, wv, ins):
        self._buf_store(wv, ins, 8)

    def i_buffer_store_dword(self, wv, ins):
        self._buf_store(wv, ins, 4)

    def i_buffer_store_byte(self, wv, ins):
        data, vaddr_o, srd_o, soff_o = ins.ops
        base, off, soff, nrec = self._buf_addr(wv, ins, vaddr_o, srd_o, soff_o)
        vals = self.vval(wv, data, ins)
        for l in np.nonzero(self.lanes(wv))[0]:
            o = int(off[l])
            if o + 1 > nrec - soff or o < 0:
                continue
            self.mem.write(base + o + soff, np.array([int(vals[l]) & 0xFF], dtype=np.uint8))
        wv.vm_ops.append({"kind": "store"})


def pack_kernarg(fields):
    """fields: list of ('q', value) / ('I', value) / ('f', value) -> bytes"""
    out = b""
    for fmt, val in fields:
        if fmt == "q":
            while len(out) % 8:
                out += b"\0"
        out += struct.pack("<" + {"q": "Q", "I": "I", "f": "f", "i": "i"}[fmt], val)
    return out
