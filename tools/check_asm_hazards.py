"""Static check of the trunk kernels' ISA: an inline-asm ``ds_read_b128`` is invisible to hipcc, which may
therefore READ (copy, at a loop back-edge or a branch join) the destination registers before the
hand-counted ``s_waitcnt lgkmcnt`` that makes the data valid.  This walks every k_trunk_x16 kernel of a
device assembly in file order, tracks the destinations of asm ds_reads that are still in flight
(retired oldest-first by the lgkmcnt immediates) and reports every instruction that reads or overwrites one.
hipcc ... -S --offload-device-only csrc/api.hip -o api.s ; python tools/check_asm_hazards.py api.s"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1)), tok[0]
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {int(m.group(2))}, m.group(1)
    return set(), None


def check(lines, name):
    pending = []                 # [(set of vgprs, line no)] oldest first
    in_asm = False
    bad = []
    for n, raw in enumerate(lines):
        l = raw.split(";")[0].strip()
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.replace(",", " ").split()]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            continue
        if in_asm and op == "ds_read_b128":
            d, _ = regs(toks[0])
            pending.append((d, n))
            continue
        if op.startswith("ds_read") or op.startswith("ds_write") or op.startswith("s_load"):
            # a compiler-visible LDS / SMEM op also counts in lgkmcnt: it sits in the queue like the asm reads
            pending.append((set(), n))
        stores = op.startswith(("ds_write", "global_store", "buffer_store", "global_load_lds", "s_"))
        srcs = toks if stores else toks[1:]
        for t in srcs:
            r, kind = regs(t)
            if kind != "v":
                continue
            for d, at in pending:
                if r & d:
                    bad.append((n, l + "   [READS]", at))
        if not stores and toks:                  # ... or overwrites one: the late LDS return would clobber the new value
            r, kind = regs(toks[0])
            if kind == "v":
                for d, at in pending:
                    if r & d:
                        bad.append((n, l + "   [OVERWRITES]", at))
    return bad


def main(path):
    L = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if l.startswith("_ZN9crl_tower11k_trunk_x16")]
    total = 0
    for k, (i, name) in enumerate(starts):
        j = starts[k + 1][0] if k + 1 < len(starts) else len(L)
        seg = L[i:j]
        end = [n for n, l in enumerate(seg) if "s_endpgm" in l]
        seg = seg[:end[-1] + 1] if end else seg
        bad = check(seg, name)
        tmpl = re.search(r"k_trunk_x16I(.*?)EEv", name).group(1).replace("Li", "").replace("E", ",").rstrip(",")
        print("k_trunk_x16<%s>: %s" % (tmpl, "ok" if not bad else "%d reads of registers with a ds_read in flight" % len(bad)))
        for n, l, at in bad[:6]:
            print("      line %d: %s   (ds_read issued at line %d)" % (n, l[:70], at))
        total += len(bad)
    return total


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
