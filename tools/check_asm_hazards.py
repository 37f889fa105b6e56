"""Static check of the trunk kernels' ISA: an inline-asm ``ds_read_b128`` is invisible to hipcc, which may
therefore READ (copy: phi moves at a loop back-edge or a branch join) its destination registers before the hand-counted ``s_waitcnt lgkmcnt`` that makes the data valid.  The
results then depend on LDS latency: right when the kernel is alone on the GPU, wrong when something delays
the LDS returns (round 3: the 64-filter split-precision kernels, found as a flaky two-rank test).

For every k_trunk_x16 kernel of a device assembly this walks the control-flow graph path by path,
carrying the destinations of the asm ds_reads still in flight (retired oldest-first by the lgkmcnt
immediates; a compiler-visible LDS / SMEM operation occupies a queue slot as well) and the scalar
constants hipcc materialises for branches it did not fold (``s_mov_b64 sX, 0`` ... ``s_and_b64 vcc, exec,
sX`` ... ``s_cbranch_vccz``: only the feasible edge is followed), and reports every instruction that
reads a register with a read in flight.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --offload-device-only \\
          chessrl_amd/csrc/api.hip -o api.s ;  python tools/check_asm_hazards.py api.s
"""
import re
import sys

MAX_STATES = 400000


def regs(tok):
    m = re.fullmatch(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1)), tok[0]
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return frozenset([int(m.group(2))]), m.group(1)
    return frozenset(), None


def parse(lines):
    """[(line no, op, operand tokens, inside inline asm)] and label -> instruction index."""
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
        ins.append((n, op, [t.strip() for t in rest.replace(",", " ").split()], in_asm, l))
    return ins, labels


def check(lines):
    ins, labels = parse(lines)
    bad = {}
    seen = set()
    # state: (instruction index, pending tuple of frozensets oldest first, constants tuple of (sgpr pair, 0 / -1), vcc knowledge)
    stack = [(0, (), (), None)]
    while stack:
        pc, pending, consts, vcc = stack.pop()
        while pc < len(ins):
            key = (pc, pending, consts, vcc)
            if key in seen:
                break
            seen.add(key)
            if len(seen) > MAX_STATES:
                raise RuntimeError("state explosion")
            n, op, toks, in_asm, text = ins[pc]
            # a materialised branch constant is used within a few instructions; forgetting it after
            # that keeps the number of distinct states small
            cd = {k: v for k, v in consts if v[1] > pc}
            consts = tuple(sorted(cd.items()))
            if len(pending) > 12:
                pending = pending[-12:]
            if op == "s_endpgm":
                break
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", text)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep else ()
                pc += 1
                continue
            if op == "s_branch":
                pc = labels[toks[0]]
                continue
            if op.startswith("s_cbranch"):
                target = labels[toks[0]]
                taken = None
                if op in ("s_cbranch_vccz", "s_cbranch_vccnz") and isinstance(vcc, tuple):
                    # vcc = exec & sX (or exec & ~sX) with sX an unknown wave-uniform flag: both edges, each
                    # REMEMBERING what it assumed about sX, so that a later test of the same flag (a
                    # prefetch and its wait hang on one condition) follows the consistent edge only
                    _, reg, negated = vcc
                    zero_if_taken = (op == "s_cbranch_vccz")             # vcc == 0 on the taken edge
                    flag_taken = (0 if zero_if_taken else -1) if not negated else (-1 if zero_if_taken else 0)
                    ct, cf = dict(consts), dict(consts)
                    ct[reg] = (flag_taken, 1 << 30)
                    cf[reg] = (-1 - flag_taken, 1 << 30)
                    stack.append((target, pending, tuple(sorted(ct.items())), None))
                    consts = tuple(sorted(cf.items()))
                    vcc = None
                    pc += 1
                    continue
                if op in ("s_cbranch_vccz", "s_cbranch_vccnz") and vcc is not None:
                    taken = (vcc == 0) if op == "s_cbranch_vccz" else (vcc != 0)
                if taken is None:
                    stack.append((target, pending, consts, None))
                    pc += 1
                elif taken:
                    pc = target
                else:
                    pc += 1
                vcc = None if taken is None else vcc
                continue
            # scalar constants and vcc
            if op == "s_mov_b64" and re.fullmatch(r"s\[\d+:\d+\]", toks[0]) and toks[1] in ("0", "-1"):
                cd[toks[0]] = (0 if toks[1] == "0" else -1, pc + 10)
            elif op in ("s_and_b64", "s_andn2_b64") and toks[0] == "vcc" and toks[1] == "exec" and toks[2] in cd:
                v = cd[toks[2]][0]
                vcc = (v if op == "s_and_b64" else (0 if v == -1 else -1))     # exec is all ones in these kernels' uniform branches
            elif op in ("s_and_b64", "s_andn2_b64") and toks[0] == "vcc" and toks[1] == "exec" and re.fullmatch(r"s\[\d+:\d+\]", toks[2]):
                vcc = ("sym", toks[2], op == "s_andn2_b64")
            else:
                if toks and toks[0] == "vcc":
                    vcc = None
                if toks and toks[0] in cd:
                    del cd[toks[0]]
                if op.startswith("s_") and toks and re.fullmatch(r"s\[\d+:\d+\]|s\d+", toks[0]):
                    # any write to an sgpr overlapping a tracked pair forgets it
                    w, _ = (frozenset(), None)
                    m = re.fullmatch(r"s\[(\d+):(\d+)\]", toks[0])
                    lo, hi = (int(m.group(1)), int(m.group(2))) if m else (int(toks[0][1:]), int(toks[0][1:]))
                    for k in list(cd):
                        a, b = (int(x) for x in re.fullmatch(r"s\[(\d+):(\d+)\]", k).groups())
                        if not (hi < a or lo > b):
                            del cd[k]
            consts = tuple(sorted(cd.items()))
            if in_asm and op == "ds_read_b128":
                d, _ = regs(toks[0])
                pending = pending + (d,)
                pc += 1
                continue
            if op.startswith(("ds_read", "ds_write", "s_load", "s_buffer_load")):
                pending = pending + (frozenset(),)       # occupies a slot of the lgkm queue
            stores = op.startswith(("ds_write", "global_store", "buffer_store", "global_load_lds", "s_", "v_cmp", "v_cmpx"))
            for t in (toks if stores else toks[1:]):
                r, kind = regs(t)
                if kind == "v" and any(r & d for d in pending):
                    bad.setdefault(n, text + "   [READS]")
            if not stores and toks:
                # an instruction that OVERWRITES such a register is not reported: hipcc reuses a fragment
                # register only where the prefetched value is dead, i.e. on paths on which the prefetch
                # was not issued at all (the prefetch and the later use hang on the same condition),
                # which this walk cannot tell from feasible ones; the new value is tracked instead
                r, kind = regs(toks[0])
                if kind == "v" and any(r & d for d in pending):
                    pending = tuple(d - r for d in pending)
            pc += 1
    return sorted(bad.items()), len(seen)


def main(path):
    L = open(path).read().split("\n")
    # every kernel of namespace crl_tower with hand-counted asm reads: k_trunk_x16<...> and the layer-wise kernels
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"_ZN9crl_tower\d+k_(trunk_x16|conv256|layer)[A-Za-z0-9_]*:", l)]
    total = 0
    for k, (i, name) in enumerate(starts):
        j = starts[k + 1][0] if k + 1 < len(starts) else len(L)
        seg = L[i + 1:j]
        end = [n for n, l in enumerate(seg) if "s_endpgm" in l]
        seg = seg[:end[-1] + 1] if end else seg
        bad, states = check(seg)
        m = re.search(r"\d+(k_[a-z0-9_]+?)I(.*?)EEv", name)
        if m is None:                                        # (not a template: no hand-counted pipeline, e.g. k_layer_touch)
            continue
        tmpl = m.group(2).replace("Li", "").replace("E", ",").rstrip(",")
        print("%s<%s>: %s" % (m.group(1), tmpl, "ok" if not bad else "%d instructions touch a register with a ds_read in flight" % len(bad)))
        for n, l in bad[:8]:
            print("      line %d: %s" % (n, l[:90]))
        total += len(bad)
    return total


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
