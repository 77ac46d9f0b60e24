"""Static check of the hand-written DPP instructions in a gfx950 assembly listing (hipcc --save-temps): the compiler's
hazard recogniser does not look inside inline asm, so this walks back from every *_dpp line and fails if a VALU
instruction writes the DPP source register less than two wait states before it, or writes EXEC (v_cmpx*, a VALU
destination of exec) less than five before it (s_nop k counts k + 1)."""
import re
import sys


def regs(op):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


def main(path):
    lines = [l.strip() for l in open(path)]
    ins = [l for l in lines if l and not l.startswith((".", ";", "//")) and not l.endswith(":")]
    bad = 0
    n = 0
    for k, l in enumerate(ins):
        if "_dpp" not in l.split()[0]:
            continue
        n += 1
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        src = regs(ops[1].split()[0])
        waits = 0
        j = k - 1
        while j >= 0 and waits < 5:
            p = ins[j]
            name = p.split()[0]
            if name == "s_nop":
                waits += int(p.split()[1], 0) + 1
            else:
                if name.startswith("v_"):
                    dst = p.split(None, 1)[1].split(",")[0].strip() if len(p.split(None, 1)) > 1 else ""
                    if waits < 2 and regs(dst) & src:
                        print("HAZARD (source written):", p, "->", l)
                        bad += 1
                    if name.startswith("v_cmpx") or dst.startswith("exec"):
                        print("HAZARD (EXEC written):", p, "->", l)
                        bad += 1
                waits += 1
            j -= 1
    print(f"{n} DPP instructions checked, {bad} hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
