"""Bisection aid for the co-residency hazard of the SLP-vectorised STFT / iSTFT kernels (NOTEBOOK R4.4): rewrite chosen
classes of packed-fp32 instructions of a gfx950 device listing (hipcc -S --cuda-device-only) into pairs of scalar VOP3 instructions with the
same operands and modifiers, leaving every other instruction - and with it the schedule around them - as it is.

    python tools/vpk_scalarize.py in.s out.s add,mul,fma [--only-kernel SUBSTR] [--keep-first N] [--keep-mod MOD] [--select plain|modified] [--if REGEX]

classes: add, mul, fma (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32).  Packed semantics (VOP3P, 64-bit operands):
    D.lo = op(S_i[op_sel[i] ? hi : lo], neg_lo[i]);   D.hi = op(S_i[op_sel_hi[i] ? hi : lo], neg_hi[i]);   op_sel = 0..., op_sel_hi = 1... by default.
An instruction whose two halves cannot be ordered without one overwriting a source of the other, or whose scalar form would read two different
SGPRs, is kept packed and counted."""
import re
import sys

PK = re.compile(r"^(\s*)v_pk_(add|mul|fma)_f32\s+(.*)$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]")


def parse_operand(tok):
    tok = tok.strip()
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return (m.group(1), int(m.group(2)))
    if re.fullmatch(r"-?\d+(\.\d+)?|0x[0-9a-fA-F]+", tok):
        return ("c", tok)          # inline constant: the same value for both halves
    return None


def half(op, hi):
    if op[0] == "c":
        return op[1]
    return "%s%d" % (op[0], op[1] + (1 if hi else 0))


def translate(indent, kind, rest, stats):
    mods = {k: [int(x) for x in v.split(",")] for k, v in MOD.findall(rest)}
    ops_txt = MOD.sub("", rest).split(";")[0].strip()
    toks = [t for t in (x.strip() for x in ops_txt.split(",")) if t]
    n_src = 3 if kind == "fma" else 2
    if len(toks) != 1 + n_src:
        stats["kept_form"] += 1
        return None
    ops = [parse_operand(t) for t in toks]
    if any(o is None for o in ops) or ops[0][0] != "v":
        stats["kept_form"] += 1
        return None
    dst, srcs = ops[0], ops[1:]
    op_sel = (mods.get("op_sel", []) + [0] * 3)[:n_src]
    op_sel_hi = (mods.get("op_sel_hi", []) + [1] * 3)[:n_src] if "op_sel_hi" in mods else [1] * n_src
    neg_lo = (mods.get("neg_lo", []) + [0] * 3)[:n_src]
    neg_hi = (mods.get("neg_hi", []) + [0] * 3)[:n_src]
    lo_src = [half(s, op_sel[i]) for i, s in enumerate(srcs)]
    hi_src = [half(s, op_sel_hi[i]) for i, s in enumerate(srcs)]
    for group in (lo_src, hi_src):
        if len({r for r in group if r.startswith("s")}) > 1:
            stats["kept_sgpr"] += 1
            return None
    d_lo, d_hi = half(dst, 0), half(dst, 1)
    name = {"add": "v_add_f32_e64", "mul": "v_mul_f32_e64", "fma": "v_fma_f32"}[kind]
    lo = "%s%s %s, %s" % (indent, name, d_lo, ", ".join(("-" if neg_lo[i] else "") + r for i, r in enumerate(lo_src)))
    hi = "%s%s %s, %s" % (indent, name, d_hi, ", ".join(("-" if neg_hi[i] else "") + r for i, r in enumerate(hi_src)))
    if d_lo not in hi_src:
        stats["done"] += 1
        return [lo, hi]
    if d_hi not in lo_src:
        stats["done"] += 1
        return [hi, lo]
    # D = op(..., swap(D), ...): exactly one source is the destination pair read crosswise - swap the pair in place first, then read it straight
    cross = [i for i, s_ in enumerate(srcs) if s_ == dst and op_sel[i] == 1 and op_sel_hi[i] == 0]
    others = [i for i, s_ in enumerate(srcs) if s_ == dst and i not in cross]
    if len(cross) == 1 and not others:
        i = cross[0]
        lo_src[i], hi_src[i] = d_lo, d_hi
        lo = "%s%s %s, %s" % (indent, name, d_lo, ", ".join(("-" if neg_lo[j] else "") + r for j, r in enumerate(lo_src)))
        hi = "%s%s %s, %s" % (indent, name, d_hi, ", ".join(("-" if neg_hi[j] else "") + r for j, r in enumerate(hi_src)))
        stats["done"] += 1
        stats["swapped"] = stats.get("swapped", 0) + 1
        return ["%sv_swap_b32 %s, %s" % (indent, d_lo, d_hi), lo, hi]
    stats["kept_overlap"] += 1
    return None


def main():
    src, dst, classes = sys.argv[1], sys.argv[2], set(sys.argv[3].split(","))
    only = None
    keep_first = 0
    keep_mod = None
    select = None
    cond = None
    nop = None
    a = sys.argv[4:]
    while a:
        if a[0] == "--only-kernel":
            only = a[1]
        elif a[0] == "--keep-first":
            keep_first = int(a[1])
        elif a[0] == "--keep-mod":          # leave packed every instruction that carries this modifier text (e.g. "neg_lo", "op_sel:")
            keep_mod = a[1]
        elif a[0] == "--if":                # scalarize only the instructions whose text matches this regular expression
            cond = re.compile(a[1])
        elif a[0] == "--nop":               # do not scalarize: put `s_nop N` in front of the selected instructions instead (they stay packed)
            nop = int(a[1])
        elif a[0] == "--select":            # scalarize only: "plain" = instructions without any modifier, "modified" = with op_sel / op_sel_hi / neg_*
            select = a[1]
        a = a[2:]
    stats = {"done": 0, "kept_form": 0, "kept_sgpr": 0, "kept_overlap": 0, "kept_by_request": 0, "other_class": 0}
    out = []
    kernel = ""
    seen = 0
    for line in open(src):
        line = line.rstrip("\n")
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
        p = PK.match(line)
        if not p:
            out.append(line)
            continue
        if p.group(2) not in classes or (only and only not in kernel):
            stats["other_class"] += 1
            out.append(line)
            continue
        seen += 1
        has_mod = bool(MOD.search(line))
        if seen <= keep_first or (keep_mod and keep_mod in line) or (select == "plain" and has_mod) or (select == "modified" and not has_mod) or (cond and not cond.search(line)):
            stats["kept_by_request"] += 1
            out.append(line)
            continue
        if nop is not None:
            stats["done"] += 1
            out.extend(["%ss_nop %d" % (p.group(1), nop), line])
            continue
        t = translate(p.group(1), p.group(2), p.group(3), stats)
        out.extend(t if t else [line])
    open(dst, "w").write("\n".join(out) + "\n")
    print("%s -> %s, classes %s: %s" % (src, dst, ",".join(sorted(classes)), stats))


if __name__ == "__main__":
    main()
